import os, sys, subprocess
for ab in (0, 2, 4, 6):
    env = dict(os.environ, XEQ_ABLATE=str(ab))
    out = subprocess.run([sys.executable, "scratch/bench_msg.py", "qm9"], env=env, capture_output=True, text=True).stdout
    print("ablate", ab, [l for l in out.splitlines() if l.startswith("mfma")][-1])
