import sys; sys.path.insert(0, '.')
import torch
from face_crop_plus_amd import engine as E
dev = torch.device('cuda:0')
x = torch.arange(2*3*4*64, dtype=torch.float32).reshape(2,3,4,64) * 0.25
a = E.Act(x.to(dev))
sp = E.f32_to_split32(a)
back = E.split32_to_f32(sp).buf.cpu()
d = (back - x).abs()
print('max err', d.max().item())
bad = (d > 1e-3).nonzero()
print('n bad', len(bad), bad[:10].tolist())
if len(bad):
    i = tuple(bad[0].tolist()); print('expected', x[i].item(), 'got', back[i].item())
raw = sp.buf.cpu().view(torch.int16).reshape(2,3,4,128)
print(raw[0,0,0,:8].tolist(), raw[0,0,1,:8].tolist())
