import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
print("torch cuda:", torch.cuda.is_available(), torch.version.hip)
def maps(tag):
    s = set()
    for line in open("/proc/self/maps"):
        if "amdhip" in line or "hsa-runtime" in line:
            s.add(line.split()[-1])
    print(tag, sorted(s))
maps("after torch")
x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
maps("after cuda init")
import gw_whisper_amd
lib = gw_whisper_amd.lib()
maps("after libgww")
h = C.c_void_p()
rc = lib.gww_frontend_create(C.byref(h))
print("frontend_create rc", rc, lib.gww_last_error())
for name in ["/opt/rocm/lib/libamdhip64.so.7", os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")]:
    try:
        r = C.CDLL(name)
        n = C.c_int(-1)
        e = r.hipGetDeviceCount(C.byref(n))
        print(name, "hipGetDeviceCount ->", e, n.value)
    except Exception as ex:
        print(name, "ERR", ex)
print({k: v for k, v in os.environ.items() if "HIP" in k or "ROC" in k or "HSA" in k or "LD_" in k})
