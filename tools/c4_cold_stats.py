import os, sys
sys.path.insert(0, "/root/repo")
os.environ["VELO_DEBUG_SKIP"] = sys.argv[1]; os.environ["VELO_ASSOC_VARIANT"] = "5"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
import velo_amd
from velo_amd import api, synth
d = synth.scan_to_map(2_000_000)
c = api.Context(0, icp_skip=1)
c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
c.associate(d["x0"], 1)
c.close()
