import ctypes as C
import torch
torch.zeros(1, device='cuda')
hip = C.CDLL([ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln][0])
for name, s in (('torch.cuda.Stream()', torch.cuda.Stream()), ('torch.cuda.Stream(priority=-1)', torch.cuda.Stream(priority=-1)), ('current', torch.cuda.current_stream())):
    fl = C.c_uint(99)
    rc = hip.hipStreamGetFlags(C.c_void_p(s.cuda_stream), C.byref(fl))
    print(name, 'handle', hex(s.cuda_stream), 'rc', rc, 'flags', fl.value, '(1 = hipStreamNonBlocking)')
