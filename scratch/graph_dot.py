"""Which nodes of the captured train step are memcpy nodes (__amd_rocclr_copyBuffer in the kernel trace), and what sits next
to them?  Captures the step with the graph's debug mode on and dumps it with hipGraphDebugDotPrint (torch: debug_dump)."""
import copy, os, re, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from preset_gen_vae_amd import config
from preset_gen_vae_amd.model import build as mbuild
from preset_gen_vae_amd.train_step import VAETrainStep
arch = sys.argv[1] if len(sys.argv) > 1 else 'speccnn4l1_bn'
B = 32
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture, mc.dim_z, mc.input_tensor_size = arch, 64, (B, 1, 257, 347)
_, _, ae = mbuild.build_ae_model(mc, tc)
ae = ae.cuda().train()
_Real = torch.cuda.CUDAGraph
made = []
def factory(*a, **k):
    g = _Real(*a, **k)
    g.enable_debug_mode()
    made.append(g)
    return g
torch.cuda.CUDAGraph = factory
step = VAETrainStep(ae, use_graph=True)
x = torch.randn(B, 1, 257, 347, device='cuda').clamp(-1, 1)
step.step(x)
torch.cuda.synchronize()
out = os.path.join(ROOT, 'gpurun_out', f'step_graph_{arch}.dot')
os.makedirs(os.path.dirname(out), exist_ok=True)
made[0].debug_dump(out)
txt = open(out).read()
nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
print(len(nodes), 'nodes', len(edges), 'edges')
kinds = {}
for n, l in nodes.items():
    k = l.split('\\n')[0][:60]
    kinds[k] = kinds.get(k, 0) + 1
for k, c in sorted(kinds.items(), key=lambda kv: -kv[1])[:60]:
    print(c, k)
for n, l in nodes.items():
    if 'MEMCPY' in l.upper() or 'copy' in l.lower():
        pre = [nodes.get(a, a)[:70] for a, b in edges if b == n]
        suc = [nodes.get(b, b)[:70] for a, b in edges if a == n]
        print('COPY NODE', l[:200].replace('\\n', ' | '), '\n   after:', pre, '\n   before:', suc)
