"""Round-4 diagnosis, step 7: victim x aggressor at op level.

The forward-level hunt (r4_hunt3.py) pins the rare wrong result on the LayerNorm-fold CONSUMER launches (qkv / fc1) of an
engine whose ViT block 0 runs while ANOTHER engine is in its stem / first ResNetV2 stage.  Here one LN-consumer GEMM of
fc1's shape (M = 1154 = two images, N = 3072, K = 768, GELU) runs back to back on one stream into distinct output buffers
while a second stream runs an "aggressor" kernel in a loop; every output is compared with the result computed alone.
A mismatch is printed as a pattern: which rows, which columns, which 128x64 tiles.

  python tools/gpu/r4_micro.py [rounds]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import DTYPES, load_library  # noqa: E402

DEV = "cuda:0"
lib = load_library()


def ptr(t):
    return None if t is None else t.data_ptr()


def st():
    return torch.cuda.current_stream().cuda_stream


class Victim:
    def __init__(self, dtype="fp16", M=1154, N=3072, K=768, act=2, ln=True):
        g = torch.Generator().manual_seed(1)
        tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
        self.dtype, self.M, self.N, self.K, self.act, self.ln = dtype, M, N, K, act, ln
        x = torch.randn(M, K, generator=g) * 2.0 + 0.3
        self.A = x.to(tdt).to(DEV)
        xa = self.A.float().cpu()
        self.W = (torch.randn(N, K, generator=g) / K ** 0.5).to(tdt).to(DEV)
        self.bias = torch.randn(N, generator=g).to(DEV)
        stats = torch.zeros(M, 8, 2)
        for b in range(K // 128):
            blk = xa[:, b * 128:(b + 1) * 128]
            stats[:, b, 0] = blk.sum(1)
            stats[:, b, 1] = (blk * blk).sum(1)
        self.stats = stats.to(DEV).contiguous()
        self.colsum = self.W.float().sum(1).contiguous()
        self.tdt = tdt

    def launch(self, out):
        if self.ln:
            rc = lib.dptx_op_gemm_ln(DTYPES[self.dtype], ptr(self.A), ptr(self.W), ptr(self.bias), ptr(out), self.M, self.N, self.K,
                                     self.act, ptr(self.stats), ptr(self.colsum), self.K // 128, 1e-6, st())
        else:
            rc = lib.dptx_op_gemm(DTYPES[self.dtype], ptr(self.A), ptr(self.W), ptr(self.bias), None, ptr(out), self.M, self.N, self.K,
                                  self.act, 0, 0, 0, st())
        assert rc == 0, rc

    def empty(self):
        return torch.empty(self.M, self.N, dtype=self.tdt, device=DEV)


def aggressor_stem(dtype="fp16"):
    x = torch.rand(2, 3, 384, 384, device=DEV)
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    Wt = (torch.randn(64, 176) * 0.1).to(tdt).to(DEV)
    y = torch.empty(2, 192, 192, 64, dtype=tdt, device=DEV)

    def run():
        assert lib.dptx_op_stem_conv(DTYPES[dtype], ptr(x), ptr(Wt), ptr(y), 2, 384, 384, st()) == 0
    return run


def aggressor_conv(dtype="fp16", Cin=64, Cout=64, H=96, k=1):
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    X = torch.randn(2, H, H, Cin, device=DEV).to(tdt)
    Wt = (torch.randn(Cout, k, k, Cin, device=DEV) * 0.05).to(tdt)
    Y = torch.empty(2, H, H, Cout, dtype=tdt, device=DEV)

    def run():
        assert lib.dptx_op_conv(DTYPES[dtype], ptr(X), ptr(Wt), None, None, ptr(Y), 2, H, H, Cin, Cout, k, 1, k // 2, k // 2, H, H, 0, 0, st()) == 0
    return run


def aggressor_torch():
    a = torch.randn(2048, 2048, device=DEV)

    def run():
        (a * 1.0001).sum()
    return run


def experiment(name, victim, aggr, rounds, per_round=24, aggr_per_round=60):
    ref = victim.empty()
    victim.launch(ref)
    torch.cuda.synchronize()
    outs = [victim.empty() for _ in range(per_round)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    shown = 0
    for r in range(rounds):
        for o in outs:
            o.fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(s2):
            if aggr is not None:
                for _ in range(aggr_per_round):
                    aggr()
        with torch.cuda.stream(s1):
            for o in outs:
                victim.launch(o)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            if torch.equal(o, ref):
                continue
            bad += 1
            if shown < 6:
                shown += 1
                d = (o != ref)
                rows = d.any(1).nonzero().flatten()
                cols = d.any(0).nonzero().flatten()
                mx = float((o.float() - ref.float()).abs().max())
                print(f"   [{name}] round {r} launch {i}: {int(d.sum())} elements differ, rows {int(rows.min())}..{int(rows.max())} ({len(rows)}), "
                      f"cols {int(cols.min())}..{int(cols.max())} ({len(cols)}), max|d| {mx:.3e}; 128-row tiles {sorted(set((rows // 128).tolist()))}, "
                      f"64-col tiles {sorted(set((cols // 64).tolist()))[:12]}", flush=True)
    print(f"[{name}] {rounds * per_round} victim launches: {bad} differ", flush=True)
    return bad


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    if os.environ.get("MICRO_FORENSICS"):
        # fc1's tile shape without the GELU: y = (acc - mu colsum) rstd + bias can be inverted element by element.  Which operand
        # of the epilogue was wrong in the elements that differ?
        v = Victim(N=3072, act=0, ln=True)
        acc = v.A.float() @ v.W.float().t()
        sm = v.stats[:, :6, 0].sum(1)
        sq = v.stats[:, :6, 1].sum(1)
        mu = sm / v.K
        var = (sq.double() / v.K - mu.double() ** 2).clamp_min(0).float()
        rstd = torch.rsqrt(var + 1e-6)
        ref = v.empty()
        v.launch(ref)
        torch.cuda.synchronize()
        model = ((acc - mu[:, None] * v.colsum[None, :]) * rstd[:, None] + v.bias[None, :])
        print(f"host model vs kernel (alone): max|d| {float((model - ref.float()).abs().max()):.3e}")
        aggr = aggressor_stem()
        outs = [v.empty() for _ in range(24)]
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        shown = 0
        for r in range(rounds):
            with torch.cuda.stream(s2):
                for _ in range(60):
                    aggr()
            with torch.cuda.stream(s1):
                for o in outs:
                    v.launch(o)
            torch.cuda.synchronize()
            for i, o in enumerate(outs):
                if torch.equal(o, ref) or shown >= 10:
                    continue
                shown += 1
                idx = (o != ref).nonzero()
                print(f"   round {r} launch {i}: {len(idx)} elements")
                for (m, n) in idx[:6].tolist() + idx[-2:].tolist():
                    yw, yr = float(o[m, n]), float(ref[m, n])
                    a, b, c, mu_, rs_ = float(acc[m, n]), float(v.bias[n]), float(v.colsum[n]), float(mu[m]), float(rstd[m])
                    c_impl = (a - (yw - b) / rs_) / mu_
                    a_impl = (yw - b) / rs_ + mu_ * c
                    print(f"      (m={m}, n={n}: tile row {m % 128}, col {n % 64}) y wrong {yw:+.4f} right {yr:+.4f} | acc {a:+.4f} implied acc {a_impl:+.4f} | "
                          f"colsum {c:+.5f} implied colsum {c_impl:+.5f} | mu {mu_:+.5f} rstd {rs_:.5f} bias {b:+.4f} | "
                          f"neighbours' colsum {[round(float(v.colsum[k]), 5) for k in range(n - 4, n + 4)]}", flush=True)
            if shown >= 10:
                break
        sys.exit(0)
    if os.environ.get("MICRO_RECORDS"):
        # library built with -DDPTX_LN_FORENSICS: the kernel itself records the operands of every element that comes out == bias
        v = Victim(N=3072, act=0, ln=True)
        trace = torch.zeros(16 + 200 * 12, dtype=torch.float32, device=DEV)
        assert lib.dptx_debug_set_trace(ptr(trace)) == 0
        ref = v.empty()
        v.launch(ref)
        torch.cuda.synchronize()
        print("records alone:", int(trace.view(torch.int32)[0]))
        aggr = aggressor_stem()
        outs = [v.empty() for _ in range(24)]
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        bad = 0
        for r in range(rounds):
            with torch.cuda.stream(s2):
                for _ in range(60):
                    aggr()
            with torch.cuda.stream(s1):
                for o in outs:
                    v.launch(o)
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        n = int(trace.view(torch.int32)[0])
        print(f"{rounds * 24} launches, {bad} differ, {n} records")
        rec = trace[16:16 + min(n, 200) * 12].view(-1, 12).cpu()
        for q in rec[:40].tolist():
            print("   m=%d n=%d raw %.5f ln_c %.5f mu %.5f rstd %.5f bias %.5f tid %d | reread mu %.5f rstd %.5f acc %.5f | recomputed %.5f" % (
                q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]))
        lib.dptx_debug_set_trace(None)
        sys.exit(0)
    if os.environ.get("MICRO_QUICK"):   # one victim x aggressor pair: A/B of library variants (DPTX_LIB)
        tag = os.path.basename(os.environ.get("DPTX_LIB", "libdptx.so"))
        experiment(f"{tag}: fc1-ln + stem conv", Victim(ln=True), aggressor_stem(), rounds)
        experiment(f"{tag}: qkv-ln + stem conv", Victim(N=2304, act=0, ln=True), aggressor_stem(), rounds)
        sys.exit(0)
    v_ln = Victim(ln=True)
    v_plain = Victim(ln=False)
    v_qkv = Victim(N=2304, act=0, ln=True)
    experiment("fc1-ln  alone", v_ln, None, rounds)
    experiment("fc1-ln  + stem conv", v_ln, aggressor_stem(), rounds)
    experiment("fc1-ln  + conv1x1 64->64 @96", v_ln, aggressor_conv(), rounds)
    experiment("fc1-ln  + conv3x3 64->64 @96", v_ln, aggressor_conv(k=3), rounds)
    experiment("fc1-ln  + torch elementwise", v_ln, aggressor_torch(), rounds)
    experiment("fc1-plain + stem conv", v_plain, aggressor_stem(), rounds)
    experiment("qkv-ln  + stem conv", v_qkv, aggressor_stem(), rounds)
    vb = Victim(dtype="bf16", ln=True)
    experiment("fc1-ln bf16 + stem conv", vb, aggressor_stem("bf16"), rounds)
