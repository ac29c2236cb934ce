#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's whole evidence set in one lease -- tools/gpu_profile.sh for c2 / c1 / c3 / c4, the
# summarisers run right here (the raw traces and counter files of four workloads exceed what gpurun copies back), and the default
# bench line.  What comes back: gpurun_out/<TAG>_profiles/ (= the files for profiles/) and gpurun_out/<TAG>_bench.json.
#   bash tools/gpu_profile_all.sh r04
set -u
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
for w in c2 c1 c3 c4; do
    t=$TAG; [ $w != c2 ] && t=${TAG}_$w
    WORKLOAD=$w bash $R/tools/gpu_profile.sh $t > $R/gpurun_out/${t}_profile.log 2>&1
    python3 $R/tools/summarize_profiles.py $t > /dev/null 2>> $R/gpurun_out/${t}_profile.log
    python3 $R/tools/summarize_traffic.py $t > /dev/null 2>> $R/gpurun_out/${t}_profile.log
    [ $w = c2 ] && python3 $R/tools/trace_timeline.py $t > $R/profiles/${t}_timeline.txt 2>> $R/gpurun_out/${t}_profile.log
    rm -rf $R/gpurun_out/$t/trace $R/gpurun_out/$t/pmc_*        # raw files stay on the box
done
mkdir -p $R/gpurun_out/${TAG}_profiles
cp $R/profiles/${TAG}* $R/gpurun_out/${TAG}_profiles/ 2>/dev/null
cd $R && python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -2 gpurun_out/${TAG}_bench.err
du -sh gpurun_out
