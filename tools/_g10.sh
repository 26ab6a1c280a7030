export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
echo "HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY"
NCCL_DEBUG=INFO timeout 120 python - <<'P' 2>&1 | grep -v "^$" | tail -40
import ctypes as C, sys
sys.path.insert(0, ".")
from pytensor_amd import ffi
ffi.init(0)
lib = ffi.lib()
ident = (C.c_ubyte * 128)()
print("uid rc", lib.pthip_comm_unique_id(ident))
rc = lib.pthip_comm_init(1, 0, ident)
print("init rc", rc, lib.pthip_last_error())
P
