"""Exit-status probe for the rocm_smi_lib path of legion_link_counters (heap corruption at exit seen in bench.py)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "no-torch": "from legion_amd import lib, engine\nprint(engine.link_counters_ex(0, 1)['supported'])",
    "torch-then-smi": "import torch\ntorch.cuda.init()\nfrom legion_amd import engine\nprint(engine.link_counters_ex(0, 1)['supported'])",
    "torch-work-then-smi": "import torch\nx=torch.zeros(1<<20,device='cuda')\nfrom legion_amd import engine\nprint(engine.link_counters_ex(0, 1)['supported'])\ny=x+1\ntorch.cuda.synchronize()\nprint(engine.link_counters_ex(0, 1)['supported'])",
    "maps": "import torch\ntorch.cuda.init()\nfrom legion_amd import engine\nengine.link_counters_ex(0, 1)\nprint([l.split()[-1] for l in open('/proc/self/maps') if 'smi' in l and 'r-xp' in l])",
    "sysfs": "import torch\ntorch.cuda.init()\nfrom legion_amd import engine\nprint(engine.link_counters_ex(0, 2)['supported'])",
}
for name, code in CASES.items():
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(f"== {name}: rc={r.returncode} {r.stdout.strip()[-300:]!r}", flush=True)
