"""Round 4, GPU: the one-wave-per-SIMD AGPR-accumulator GEMM (csrc/gemm_v7.hip; tile variants 13 = 256x256, 14 = 256x320,
15 = 192x256) against an fp32 matmul, against the other tile variants bit for bit (same accumulation order by construction), and
against itself over repeated launches (its counted DMA waits and one barrier per k-step leave no slack for a misplaced read)."""
import math
import os

import pytest
import torch

from tiny_audio_amd import ops

pytestmark = pytest.mark.gpu
DEV, BF16, F32 = "cuda", torch.bfloat16, torch.float32
V7 = {13: (256, 256, 3), 14: (256, 320, 4), 15: (192, 256, 12)}      # variant -> (BM, BN, ping-pong variant of the same tile)


def rnd(*shape, seed=0, scale=1.0, dtype=F32):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


class variant:
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = os.environ.get("TA355_GEMM_VARIANT")
        os.environ["TA355_GEMM_VARIANT"] = str(self.v)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("TA355_GEMM_VARIANT", None)
        else:
            os.environ["TA355_GEMM_VARIANT"] = self.old


# (M, N, K): full tiles, ragged edges in both directions, fewer tiles than CUs, several tiles per workgroup (> 256 tiles),
# the shortest K the kernel serves (128 = four k-steps) and the step's own shapes
SHAPES = [(512, 512, 256), (256, 320, 128), (500, 388, 192), (1000, 1284, 128), (2048, 2048, 1024), (4000, 1024, 1024),
          (6144, 4096, 1024), (6144, 2048, 2048), (6144, 1024, 3072), (16000, 1280, 1280), (16000, 5120, 1280), (16000, 3840, 1280),
          (8192, 8192, 512)]


@pytest.mark.parametrize("v", sorted(V7))
@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in SHAPES])
def test_v7_plain_vs_fp32_and_vs_pingpong(v, M, N, K):
    A, W = rnd(M, K, seed=1, dtype=BF16), rnd(N, K, seed=2, scale=1 / math.sqrt(K), dtype=BF16)
    ref = A.float() @ W.float().T
    with variant(v):
        yb = ops.gemm_nt(A, W, out_dtype=BF16)
        yf = ops.gemm_nt(A, W, out_dtype=F32)
    assert relerr(yb, ref) < 1.5e-2 and relerr(yf, ref) < 2e-3
    with variant(V7[v][2]):
        zb = ops.gemm_nt(A, W, out_dtype=BF16)
        zf = ops.gemm_nt(A, W, out_dtype=F32)
    assert torch.equal(yb, zb) and torch.equal(yf, zf)          # k ascending, fp32, one accumulator per output: bit-identical


@pytest.mark.parametrize("v", sorted(V7))
def test_v7_epilogues(v):
    M, N, K = 3000, 1600, 640                                      # ragged in M (both tile heights) and N (both tile widths)
    A, W = rnd(M, K, seed=1, dtype=BF16), rnd(N, K, seed=2, scale=1 / math.sqrt(K), dtype=BF16)
    bias, xr = rnd(N, seed=3), rnd(M, N, seed=4, dtype=BF16)
    ref = A.float() @ W.float().T
    other = V7[v][2]
    outs = {}
    for tag, vv in (("v7", v), ("pp", other)):
        with variant(vv):
            o = {}
            o["bias_bf16"] = ops.gemm_nt(A, W, bias=bias, out_dtype=BF16)
            o["bias_gelu"] = ops.gemm_nt(A, W, bias=bias, act=1, out_dtype=BF16)
            o["res_f32"] = ops.gemm_nt(A, W, bias=bias, residual=xr.float().contiguous(), out_dtype=F32)
            o["res_f32_bf16out"] = ops.gemm_nt(A, W, residual=xr.float().contiguous(), out_dtype=BF16)
            x = xr.clone(); ops.gemm_nt(A, W, bias=bias, out=x, residual_bf16=x); o["res_bf16_inplace"] = x
            o["res_bf16"] = ops.gemm_nt(A, W, out_dtype=BF16, residual_bf16=xr)
            o["splitk"] = ops.gemm_nt(A, W, out_dtype=F32, splits=2)
            outs[tag] = o
    a = outs["v7"]
    assert relerr(a["bias_bf16"], ref + bias) < 1.5e-2
    assert relerr(a["bias_gelu"], torch.nn.functional.gelu(ref + bias)) < 1.5e-2
    assert relerr(a["res_f32"], ref + bias + xr.float()) < 2e-3
    assert relerr(a["res_f32_bf16out"], ref + xr.float()) < 1.5e-2
    assert relerr(a["res_bf16_inplace"], ref + bias + xr.float()) < 1.5e-2
    assert relerr(a["res_bf16"], ref + xr.float()) < 1.5e-2
    assert relerr(a["splitk"], ref) < 2e-3
    for k in a:
        assert torch.equal(a[k], outs["pp"][k]), k


@pytest.mark.parametrize("v", sorted(V7))
def test_v7_rope_epilogue_and_column_limit(v):
    """The encoder's q|k|v launch: rope on the first 2H columns only, M = 16 clips x 500 frames."""
    from tests.test_gpu_kernels import il_perm, rope_tables, rot_half
    S, H, nh = 500, 1280, 20
    M, N, K = 16 * S, 3 * H, H
    A, W = rnd(M, K, seed=1, dtype=BF16), rnd(N, K, seed=2, scale=1 / math.sqrt(K), dtype=BF16)
    bias = 0.1 * rnd(N, seed=3)
    cos, sin = rope_tables(1500, 32, 10000.0)
    tab = torch.stack([cos, sin], -1).contiguous()
    rows = torch.cat([(torch.arange(2 * nh)[:, None] * 64 + il_perm()[None, :]).reshape(-1), torch.arange(2 * H, 3 * H)]).to(DEV)
    Wp, bp = W[rows].contiguous(), bias[rows].contiguous()
    with variant(v):
        out = ops.gemm_nt(A, Wp, bias=bp, act=2, rope=(tab, S, 2 * H))
    with variant(V7[v][2]):
        out2 = ops.gemm_nt(A, Wp, bias=bp, act=2, rope=(tab, S, 2 * H))
    assert torch.equal(out, out2)
    ref = A.float() @ W.float().T + bias
    y = ref[:, :2 * H].reshape(M // S, S, 2 * nh, 64)
    c = torch.cat([cos[:S], cos[:S]], -1)[None, :, None]; s = torch.cat([sin[:S], sin[:S]], -1)[None, :, None]
    want = torch.cat([y[..., :32] * c + rot_half(y[..., :32]) * s, y[..., 32:]], -1)[..., il_perm().to(DEV)].reshape(M, 2 * H)
    assert relerr(out[:, :2 * H], want) < 8e-3
    assert relerr(out[:, 2 * H:], ref[:, 2 * H:]) < 8e-3


@pytest.mark.parametrize("v", sorted(V7))
def test_v7_row_mapped_conv(v):
    """conv2 of the encoder: overlapping rows of the padded time-major buffer (stride 2), GELU."""
    B, T, H = 8, 1000, 1280
    S = T // 2
    x1 = torch.zeros(B, T + 2, H, device=DEV, dtype=BF16); x1[:, 1:T + 1] = rnd(B, T, H, seed=1, dtype=BF16)
    w2, b2 = rnd(H, 3 * H, seed=4, scale=1 / math.sqrt(3 * H), dtype=BF16), rnd(H, seed=5)
    with variant(v):
        xr = ops.gemm_nt(x1, w2, M=B * S, N=H, K=3 * H, bias=b2, act=1, out_dtype=BF16, a_map=(2 * H, S, (T + 2) * H))
    win2 = torch.cat([x1[:, 0:T:2], x1[:, 1:T + 1:2], x1[:, 2:T + 2:2]], -1).float().reshape(B * S, 3 * H)
    assert relerr(xr, torch.nn.functional.gelu(win2 @ w2.float().T + b2)) < 1.5e-2
    with variant(V7[v][2]):
        xr2 = ops.gemm_nt(x1, w2, M=B * S, N=H, K=3 * H, bias=b2, act=1, out_dtype=BF16, a_map=(2 * H, S, (T + 2) * H))
    assert torch.equal(xr, xr2)


@pytest.mark.parametrize("v,M,N,K", [(13, 8192, 4096, 1024), (14, 16000, 5120, 1280), (15, 6144, 4096, 1024), (14, 16000, 1280, 5120)])
def test_v7_repeated_launches_are_identical(v, M, N, K):
    """Race hunt: 30 launches over the same operands (with a different kernel in between that evicts L2), all bit-identical."""
    A, W = rnd(M, K, seed=1, dtype=BF16), rnd(N, K, seed=2, scale=1 / math.sqrt(K), dtype=BF16)
    junk = torch.empty(64 * 1024 * 1024, device=DEV, dtype=torch.int16)
    with variant(v):
        first = ops.gemm_nt(A, W, out_dtype=BF16).clone()
        for it in range(30):
            if it % 3 == 0:
                junk.fill_(it)
            y = ops.gemm_nt(A, W, out_dtype=BF16)
            assert torch.equal(y, first), it


def test_v7_unserved_launches_fall_back():
    """K = 64 (two k-steps) and the K extension are outside the kernel: the launch must still give the right answer (fallback tile)."""
    A, W = rnd(300, 64, seed=1, dtype=BF16), rnd(256, 64, seed=2, scale=1 / 8, dtype=BF16)
    with variant(13):
        y = ops.gemm_nt(A, W, out_dtype=F32)
    assert relerr(y, A.float() @ W.float().T) < 2e-3
    M, N, K = 512, 512, 256
    A, W = rnd(M, K, seed=3, dtype=BF16), rnd(N, K, seed=4, scale=1 / 16, dtype=BF16)
    A2, W2 = rnd(M, 64, seed=5, dtype=BF16), rnd(N, 64, seed=6, scale=1 / 8, dtype=BF16)
    with variant(14):
        y = ops.gemm_nt(A, W, out_dtype=F32, k_ext=(A2, W2))
    assert relerr(y, A.float() @ W.float().T + A2.float() @ W2.float().T) < 2e-3
