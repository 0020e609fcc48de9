"""Helpers for the -m gpu tests: call libdptx.so op entry points on torch CUDA tensors."""
import torch

from omnidata_amd.engine import load_library, DTYPES

TDT = {"bf16": torch.bfloat16, "fp16": torch.float16}
# max |err| allowed relative to max |ref| when the OUTPUT is rounded to the 16-bit type
# (half an ulp of the largest value = 2^-9 / 2^-12, plus fp32 accumulation-order noise)
OUT_TOL = {"bf16": 6e-3, "fp16": 8e-4}


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def rel_err(a, ref):
    a, ref = a.double(), ref.double()
    return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


def op_gemm(dtype, A, W, bias=None, R=None, act=0, c_fp32=False):
    lib = load_library()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty(M, N, device=A.device, dtype=torch.float32 if c_fp32 else TDT[dtype])
    rc = lib.dptx_op_gemm(DTYPES[dtype], ptr(A), ptr(W), ptr(bias), ptr(R), ptr(C), M, N, K, act,
                          int(A.dtype == torch.float32), int(c_fp32), int(R is not None and R.dtype == torch.float32), stream())
    assert rc == 0, rc
    return C


def op_conv(dtype, X, Wt, bias, R, stride, pad_t, pad_l, Ho, Wo, a_relu=0, act=0):
    """X NHWC [B,H,W,Cin]; Wt [Cout,k,k,Cin]."""
    lib = load_library()
    B, H, W, Cin = X.shape
    Cout, k = Wt.shape[0], Wt.shape[1]
    Y = torch.empty(B, Ho, Wo, Cout, device=X.device, dtype=TDT[dtype])
    rc = lib.dptx_op_conv(DTYPES[dtype], ptr(X), ptr(Wt), ptr(bias), ptr(R), ptr(Y), B, H, W, Cin, Cout, k, stride,
                          pad_t, pad_l, Ho, Wo, a_relu, act, stream())
    assert rc == 0, rc
    return Y
