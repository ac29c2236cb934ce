"""Dev tools that use the A/B / diagnostic environment switches import this FIRST: the product library reads five documented
knobs only (include/velo_hip.h, "Environment"), every other switch exists in the -DVELO_DIAGNOSTICS build -- which is what
api.load_library() then loads for the whole process (VELO_LIB_PATH)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import velo_amd  # noqa: E402,F401
from velo_amd import build  # noqa: E402

if "VELO_LIB_PATH" not in os.environ:
    os.environ["VELO_LIB_PATH"] = build.build_hip(diagnostics=True)   # prebuilt in the authoring container; a no-op on the GPU box
