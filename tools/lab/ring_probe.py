import sys, torch
sys.path.insert(0, '.')
from ralf_amd.helpers import task
img = torch.rand(64, 3, 256, 256); sal = torch.rand(64, 1, 256, 256)
for i in range(10):
    a = task.cat_image(img, sal)
    ring = task._PINNED[(64, 4, 256, 256, torch.float32)]["bufs"]
    print(i, a.is_pinned(), len(ring), [torch._C._storage_Use_Count(s[0].untyped_storage()._cdata) for s in ring])
    del a
