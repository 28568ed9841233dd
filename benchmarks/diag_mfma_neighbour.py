"""Diagnostic: does an MFMA-dense kernel on another stream change the results of simple known-answer
kernels (one instruction class each)?  Uses the test-only kernels of tests/hip."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = C.CDLL(os.path.join(ROOT, "tests", "hip", "libqz_testkernels.so"))
T.qzt_stress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
T.qzt_victim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
sout = torch.zeros(2048 * 256, dtype=torch.float32, device=dev)
sin = torch.randn(2048 * 256 + 16 * 65536, dtype=torch.float32, device=dev)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
NB = 683
for kind, name in ((0, "v_pk_fma_f32"), (1, "v_fma_f32"), (2, "exp/tanh"), (3, "LDS round trips"), (4, "ds_bpermute shuffles")):
    ref = torch.zeros(NB * 192, dtype=torch.float32, device=dev)
    T.qzt_victim(ref.data_ptr(), NB, 2000, kind, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for neighbour, mode in (("nothing", None), ("MFMA loop", 1), ("VALU-only loop", 0)):
        bad = 0
        for rep in range(50):
            out = torch.zeros_like(ref)
            torch.cuda.synchronize()
            if mode is not None:
                T.qzt_stress(sout.data_ptr(), sin.data_ptr(), 2048, 300, mode, 1, B.cuda_stream)
            T.qzt_victim(out.data_ptr(), NB, 2000, kind, A.cuda_stream)
            if mode is not None:
                T.qzt_stress(sout.data_ptr(), sin.data_ptr(), 2048, 300, mode, 1, B.cuda_stream)
            torch.cuda.synchronize()
            bad += int(not torch.equal(out, ref))
        print("victim %-22s next to %-15s: %d of 50 runs differ" % (name, neighbour, bad), flush=True)
