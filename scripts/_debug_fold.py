import sys, torch
sys.path.insert(0, '.')
import petr_amd
from tests import test_neck as T
from oracle import neck_oracle as NO
from oracle import petr_oracle as O
T.test_cpfpn_into_head_training_step_gpu()
torch.cuda.synchronize(); print('into_head ok', flush=True)
import gc; gc.collect(); torch.cuda.synchronize(); print('gc ok', flush=True)
torch.manual_seed(5)
chans, sizes = [768, 1024], [(40, 100), (20, 50)]
neck = petr_amd.build_neck(dict(type='CPFPN', in_channels=chans, out_channels=256, num_outs=2)); neck.init_weights()
head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=100))
g = torch.Generator().manual_seed(11)
inputs = [torch.randn(6, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]
with torch.no_grad():
    sd64 = {k: v.double() for k, v in neck.state_dict().items()}
    lvl0 = NO.cpfpn_forward(sd64, [x.double() for x in inputs])[0]
    print('oracle neck ok', flush=True)
    want_mem = torch.nn.functional.conv2d(lvl0, head.input_proj.weight.double(), head.input_proj.bias.double()).permute(0, 2, 3, 1)
print('oracle ok', flush=True)
neck, head = neck.cuda().eval(), head.cuda().eval()
xs = [x.cuda() for x in inputs]
torch.cuda.synchronize(); print('setup ok', flush=True)
wf, bf = neck._folded_w3(head)
torch.cuda.synchronize(); print('fold ok', flush=True)
with torch.no_grad():
    lat, pad = neck._laterals_topdown(xs)
torch.cuda.synchronize(); print('laterals ok', pad.shape, flush=True)
mem = neck.forward_folded(xs, head)
torch.cuda.synchronize(); print('folded ok', flush=True)
m = mem.cpu()
print('copied', flush=True)
d = m.double()
print('double', flush=True)
e = d - want_mem
print('sub', flush=True)
print(e.abs().max().item())
