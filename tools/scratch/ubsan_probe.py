"""Why do the conv2 launches of the host-UBSan library write nothing?  (run with DTF_KERNELS_LIB=...)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import native
lib = native.load()
hip = ctypes.CDLL("libamdhip64.so")
B, C1, C2 = 4, 32, 64
x = torch.rand(B, 14, 14, C1, device="cuda").bfloat16()
w = (torch.randn(25 * C1 * C2, device="cuda") * 0.03).bfloat16()
packed = torch.empty(2, 25 * C1 * C2, dtype=torch.bfloat16, device="cuda")
b = torch.zeros(C2, device="cuda")
pooled = torch.full((B, 7, 7, C2), -7.0, dtype=torch.bfloat16, device="cuda")
arg = torch.full((B, 7, 7, C2), 9, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
print("pack rc", lib.dtf_conv2_pack_weights(w.data_ptr(), packed.data_ptr(), C1, C2, s))
torch.cuda.synchronize()
os.environ["AMD_LOG_LEVEL"] = "0"
rc = lib.dtf_conv2_fwd(x.data_ptr(), packed.data_ptr(), b.data_ptr(), pooled.data_ptr(), arg.data_ptr(), B, C1, C2, s)
print("fwd rc", rc, "last error", hip.hipGetLastError())
try:
    torch.cuda.synchronize()
    print("sync ok")
except Exception as e:
    print("sync raised", e)
print("pooled untouched:", bool((pooled == -7.0).all()), "arg untouched:", bool((arg == 9).all()), "max", float(pooled.float().max()))
