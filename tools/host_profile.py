"""Where does the host spend its time while enqueuing a training step? cProfile over N eager steps (GPU box).
usage: python tools/host_profile.py [unet|deeplab] [steps]"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import insar_unet_ca_amd as iu
from insar_unet_ca_amd.data import make_batch

which = sys.argv[1] if len(sys.argv) > 1 else "unet"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
if which == "unet":
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train(); crit = iu.DiceCELoss(ignore_index=255); ch = 2
else:
    net = iu.DeepLabV3_SingleChannel_Attn(2, "resnet50", False, compute_dtype=torch.bfloat16).to(dev).train(); crit = iu.CrossEntropyLoss(ignore_index=255); ch = 1
opt = iu.Adam(net.parameters(), lr=1e-4)
x, y = (t.to(dev) for t in make_batch(0, 16, 256, channels=ch))


def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(net(x), y); loss.backward(); opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(22)
print(s.getvalue()[:6000])
