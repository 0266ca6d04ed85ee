"""Build attention.hip with the given -D settings (e.g. GWW_ATT_EPI16=0 GWW_ATT_EPI16=1), link each against the prebuilt
objects and time the bench-shape attention in a child process per build (run on the GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "att_ab"); os.makedirs(out, exist_ok=True)
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f != "attention.o"]
for n, v in enumerate(sys.argv[1:]):
    o = os.path.join(out, f"attention_{n}.o"); so = os.path.join(out, f"libgww_att{n}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                    f"-D{v}", "-c", os.path.join(csrc, "attention.hip"), "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run", "att_time.py")], env=dict(os.environ, GWW_LIB=so),
                       capture_output=True, text=True)
    print(f"{v}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ''} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
