"""rocprofv3 helper: the GRU Scan (config #5) at T=32 so the kernel trace stays small."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc
from pytensor_amd import configs, ffi
ffi.init(0)
v = configs.c5_inputs(T=32, B=64, H=1024)
td, tw = bc.run_case("c5_gru", v, 5, check=False)
print({"T": 32, "ms_device": td, "ms_per_step": td / 32})
