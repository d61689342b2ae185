"""Micro-benchmark: EfficientTrack forward (N images 256x256) with per-launch HIP-event timing."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from jarvis_hybridnet_amd import _native as N, synthetic as S
from jarvis_hybridnet_amd.efficienttrack.model import EfficientTrackBackbone
NI = int(os.environ.get('N', '96'))
net = EfficientTrackBackbone(None, 'small', 23); net.load_state_dict(S.efficienttrack_weights('small', 23, 4))
x = torch.randn(NI, 3, 256, 256, device='cuda')
for _ in range(3): y = net(x)
torch.cuda.synchronize()
recs = []
for _ in range(5): recs += N.profile(lambda: net(x))
agg = {}
for name, ms, fl, by in recs:
    a = agg.setdefault(name, [0.0, 0, fl, by]); a[0] += ms; a[1] += 1
tot = sum(a[0] for a in agg.values())/5
print('EffTrack N=%d total %.3f ms/fwd' % (NI, tot))
for k, (ms, n, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get('TOP','14'))]:
    print('  %-28s n=%3d avg %7.1f us  tot %.3f ms  %6.1f TF/s %7.1f GB/s' % (k, n//5, 1e3*ms/n, ms/5, fl/(ms/n)/1e9 if fl else 0, by/(ms/n)/1e6))
