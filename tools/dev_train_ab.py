"""Same-box A/B of the training step (bench.py --workload config3) under an environment switch: tools/dev_train_ab.py VAR a b [dtype]"""
import json, os, subprocess, sys
var, va, vb = sys.argv[1:4]
dt = sys.argv[4] if len(sys.argv) > 4 else "bf16"
for rep in range(3):
    for v in (va, vb):
        p = subprocess.run([sys.executable, "bench.py", "--workload", "config3", "--steps", "6", "--warmup", "3", "--dtype", dt], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1]); print(f"{var}={v}: {d['ms_per_step']:.2f} ms/step  losses {d['config'].get('last_losses')}", flush=True)
        except Exception:
            print(f"{var}={v}: failed", p.stderr[-400:], flush=True)
