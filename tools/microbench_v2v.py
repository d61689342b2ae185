"""Micro-benchmark: V2VNet forward (T volumes of 64^3) with per-launch HIP-event timing."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from jarvis_hybridnet_amd import _native as N, synthetic as S
from jarvis_hybridnet_amd.hybridnet.v2vnet import V2VNet
T = int(os.environ.get('T', '8')); G = 64; J = 23
net = V2VNet(J, J); net.load_state_dict(S.v2v_weights(J, 22))
x = torch.rand(T, J, G, G, G, device='cuda')
for _ in range(3): y = net(x)
torch.cuda.synchronize()
recs = []
for _ in range(5): recs += N.profile(lambda: net(x))
agg = {}
for name, ms, fl, by in recs:
    a = agg.setdefault(name, [0.0, 0, fl]); a[0] += ms; a[1] += 1
tot = sum(a[0] for a in agg.values())/5
print('V2V T=%d total %.3f ms/fwd  (%.1f TFLOP/s overall)' % (T, tot, sum(a[2]*a[1] for a in agg.values())/5/tot/1e9))
for k, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('  %-28s n=%2d avg %.3f ms  %.1f TF/s' % (k, n//5, ms/n, fl/(ms/n)/1e9 if fl else 0))
