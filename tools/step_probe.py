"""Per-timestep forward path on a few shapes (forward us per launch): python tools/step_probe.py   (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for shape in (('128', '300', '4096'), ('512', '300', '1440'), ('64', '300', '1440')):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'phase_probe.py'), *shape, 'pruned'], capture_output=True, text=True).stdout
    print(out.strip().splitlines()[-1] if out.strip() else 'no output')
