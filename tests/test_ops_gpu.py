"""Kernel-level parity (GPU): every conv kernel of libx3dhip.so, called through the C ABI,
against a float64 CPU evaluation of the same fused op built from the oracle's per-op
functions (oracle/x3d_oracle.py).  Tolerance 2e-5 relative L2 (fp32 kernels vs fp64 truth)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import x3d_oracle as xo

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _canary_bands_around_every_output(request):
    """Every KERNEL-LEVEL test of this file runs with guard-band allocation (x3dhip.ops.set_guard): each buffer the ops layer allocates sits
    between two 4 KB canary bands, checked at teardown -- a kernel that writes outside its output at ANY of these shapes
    (odd planes, P % 4 != 0, tail tiles, strided gathers) fails the test even when its own output is right."""
    kernel_level = request.node.name.startswith(("test_pw", "test_dw333", "test_stem", "test_elementwise", "test_head",
                                                 "test_reduce", "test_three_term"))
    if not torch.cuda.is_available() or not kernel_level:       # (block / model / trainer tests measure memory and hold graphs)
        yield
        return
    from x3dhip import ops
    prev = ops.set_guard(True)
    yield
    torch.cuda.synchronize()
    bad = ops.check_guards()
    ops.set_guard(prev)
    assert not bad, "%d buffers written out of bounds, first: %s" % (len(bad), bad[:6])

TOL = 2e-5


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def _act(s, act):
    if act == 1:
        return torch.relu(s)
    if act == 2:
        return s * torch.sigmoid(s)
    return s


def _dact(s, act):
    s = s.detach().clone().requires_grad_(True)
    _act(s, act).sum().backward()
    return s.grad


def _g(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


PW_CASES = [
    # N, Cin, Cout, T, H, W, stride, act
    (2, 24, 54, 4, 14, 14, 1, 0),
    (2, 54, 24, 4, 14, 14, 1, 2),
    (1, 24, 54, 2, 9, 7, 1, 1),       # P % 4 != 0 -> scalar path
    (2, 24, 24, 4, 15, 13, 2, 1),     # strided gather, odd sizes
    (1, 216, 96, 2, 6, 6, 1, 2),      # K > 32 (7 chunks), M 96 (6 tiles)
    (1, 96, 432, 2, 5, 5, 1, 0),      # M 432 -> 4 M-blocks, scalar path
    (1, 192, 432, 4, 4, 4, 1, 0),
    (3, 48, 108, 4, 20, 20, 1, 0),    # 7 M-tiles
    (2, 96, 216, 4, 12, 12, 1, 2),    # LDS-tiled variant: 9 voxel tiles, 14 M-tiles (2 blocks)
    (2, 96, 192, 2, 9, 9, 2, 1),      # LDS-tiled variant, strided gather (layer4.0 downsample)
    (2, 432, 192, 2, 7, 7, 1, 2),     # LDS-tiled variant, K = 432, P % 4 != 0
    (1, 100, 130, 3, 6, 6, 1, 1),     # LDS-tiled, K % 4 == 0 but M not a multiple of 16
    (1, 70, 98, 2, 6, 6, 1, 0),       # LDS-tiled, K % 4 != 0 (scalar weight staging)
    (20, 96, 112, 8, 14, 14, 1, 2),   # persistent tiled kernel: 800 items > grid (2 items per workgroup), 7 M-tiles (U = 4)
    (20, 128, 96, 4, 16, 16, 1, 1),   # persistent tiled kernel: 4 chunks per item, 6 M-tiles (U = 3), item list with a tail
    (2, 48, 96, 4, 16, 16, 2, 1),     # stride 2 with P_out % 4 == 0: gathered-input weight-gradient kernel (downsample branch)
    (3, 24, 24, 2, 12, 20, 2, 0),     # same, narrow channels (64 x 32 tile), raw input
    # X3D-S eval, stage 4 at 2 x 13 x 160^2 (P = 13 * 5 * 5 = 325, P % 4 != 0): M = 432 is 27 sixteen-row tiles in 4 blocks
    # of 7, so the last block owns a tile index (27) past the packed image.  Round 1's first tiled kernel (commits
    # 48232ad..cf54da2) read that tile's fragments unclamped -- up to 12 KiB beyond the pack allocation, results never
    # stored -- which aborted tests/test_model_gpu.py::test_eval_forward_config1_S whenever the fresh pack sat at the end
    # of a mapped segment (gpurun_out/all23.log); every kernel has clamped the tile index since 3e45adf
    (2, 192, 432, 13, 5, 5, 1, 1),
    (2, 432, 192, 13, 5, 5, 1, 2),
    # the pointwise layers of X3D-M at the multigrid clip (2, 4, 158, 158) with 2 BN splits (GPUTEST_r02's red case
    # train_M_2x4x158_s2): planes 79 -> 40 -> 20 -> 10 -> 5, P = 24 964 / 6400 / 1600 / 400 / 100.  P = 100 ends in a 4-voxel
    # tile of the 32-voxel whole-K kernels; the downsample convs gather from odd planes (79 -> 40)
    (2, 24, 24, 4, 79, 79, 2, 1),     # layer1.0 downsample: strided gather from the odd 79 x 79 plane
    (2, 24, 48, 4, 40, 40, 2, 1),     # layer2.0 downsample
    (2, 48, 216, 4, 20, 20, 1, 1),    # layer3.0 conv1
    (2, 48, 96, 4, 20, 20, 2, 1),     # layer3.0 downsample
    (2, 216, 96, 4, 10, 10, 1, 2),    # layer3.x conv3 (P = 400: 13 tiles of 32, tail of 16)
    (2, 96, 216, 4, 10, 10, 1, 0),    # layer3.x conv1
    # the pointwise layers of the smallest golden clip (2, 4, 32, 32): P = 256 / 64 / 16 / 4 (stage 4: ONE voxel per plane)
    (2, 24, 54, 4, 8, 8, 1, 0), (2, 54, 24, 4, 8, 8, 1, 2), (2, 48, 108, 4, 4, 4, 1, 0), (2, 108, 48, 4, 4, 4, 1, 2),
    (2, 96, 216, 4, 2, 2, 1, 0), (2, 216, 96, 4, 2, 2, 1, 2), (2, 96, 192, 4, 2, 2, 2, 1),
    (2, 192, 432, 4, 1, 1, 1, 1), (2, 432, 192, 4, 1, 1, 1, 2),
    (2, 96, 432, 4, 10, 10, 1, 0),    # layer4.0 conv1
    (2, 96, 192, 4, 10, 10, 2, 1),    # layer4.0 downsample
    (2, 432, 192, 4, 5, 5, 1, 2),     # layer4.x conv3 (P = 100: 4 tiles of 32, tail of 4)
    (2, 192, 432, 4, 5, 5, 1, 0),     # layer4.x conv1 / conv5
]


@pytest.mark.parametrize("case", PW_CASES)
def test_pw_fwd(case):
    from x3dhip import ops
    dev = _dev()
    N, Ci, Co, T, H, W, s, act = case
    x = _g(N, Ci, T, H, W, seed=1)
    w = _g(Co, Ci, seed=2) / np.sqrt(Ci)
    pre = torch.stack([1 + 0.2 * _g(N, Ci, seed=3), 0.3 * _g(N, Ci, seed=4)], -1) if act else None
    xin = _act(pre[..., 0, None, None, None] * x + pre[..., 1, None, None, None], act) if act else x
    y_ref = xo.pw(xin, w.view(Co, Ci, 1, 1, 1), s)
    wd = w.float().to(dev)
    for wp in (None, ops.pw_pack(wd)):          # streaming kernel, then (large layers) the tiled one
        y, partial = ops.pw_fwd(x.float().to(dev), wd, stride=s,
                                pre=None if pre is None else pre.float().contiguous().to(dev), pre_act=act, wp=wp)
        assert _rel(y, y_ref) < TOL
        st = partial.double().sum(2).cpu()
        assert _rel(st[..., 0], y_ref.sum(dim=(2, 3, 4))) < 5 * TOL + 1e-6
        assert _rel(st[..., 1], (y_ref ** 2).sum(dim=(2, 3, 4))) < 5 * TOL


# (N, Cin, Cout, T, H, W, act): the stage 3-4 layers the persistent producer / consumer kernel takes (K <= 224), with tail
# tiles (P % 32 != 0), M not a multiple of 16, K not a multiple of 32, two M blocks (27 tiles), one tile per wave and two
P8_CASES = [(8, 96, 216, 16, 14, 14, 0), (8, 216, 96, 16, 14, 14, 2), (8, 192, 432, 16, 7, 7, 0), (2, 96, 216, 4, 10, 10, 1),
            (2, 216, 96, 4, 10, 10, 2), (3, 72, 162, 4, 10, 10, 0), (2, 162, 100, 5, 6, 6, 2), (2, 192, 432, 4, 5, 5, 0),
            (1, 64, 96, 1, 2, 2, 1), (2, 128, 280, 4, 7, 7, 2), (5, 200, 120, 3, 6, 10, 0)]


@pytest.mark.parametrize("case", P8_CASES)
@pytest.mark.parametrize("grid", [0, 3, 40])
def test_pw_fwd_persistent_kernel_is_bitwise_the_whole_k_kernel(case, grid):
    """pw8_kernel (round 4: persistent, producer / consumer waves, double-buffered LDS, resident A fragments) against
    pw6_kernel (option no_pw8) on the same inputs: y and the statistics tiles BITWISE equal (same products, same
    accumulation order), for the default grid (one workgroup per CU) and for 3 / 40 workgroups (up to hundreds of items
    per workgroup: the item loop, the buffer parity and the odd / even iteration tails)."""
    from x3dhip import _lib, ops
    dev = _dev()
    N, Ci, Co, T, H, W, act = case
    x = _g(N, Ci, T, H, W, seed=1).float().to(dev)
    w = (_g(Co, Ci, seed=2) / np.sqrt(Ci)).float().to(dev)
    pre = torch.stack([1 + 0.2 * _g(N, Ci, seed=3), 0.3 * _g(N, Ci, seed=4)], -1).float().contiguous().to(dev) if act else None
    wp = ops.pw_pack(w)
    with _lib.options(no_pw8=1):
        y0, p0 = ops.pw_fwd(x, w, pre=pre, pre_act=act, wp=wp)
        k0 = _lib.last_kernel()
    with _lib.options(pw8_grid=grid, pw8_max_k=224):
        y1, p1 = ops.pw_fwd(x, w, pre=pre, pre_act=act, wp=wp)
        k1 = _lib.last_kernel()
    torch.cuda.synchronize()
    assert k0 == "pw6_kernel" and k1 == "pw8_kernel", (k0, k1)
    assert torch.equal(y1, y0)
    assert torch.equal(p1, p0)


@pytest.mark.parametrize("case", [(8, 432, 192, 16, 7, 7), (2, 432, 192, 4, 5, 5), (2, 432, 200, 3, 6, 6), (3, 352, 150, 2, 5, 8),
                                  (2, 448, 256, 2, 4, 4),
                                  # mode 2: more than 8 M tiles at any K (27 tiles = two per wave, 14 tiles, 32 tiles)
                                  (2, 192, 432, 4, 5, 5), (2, 96, 216, 4, 10, 10), (2, 128, 512, 2, 4, 4), (2, 216, 300, 3, 6, 6)])
def test_pw_sixteen_wave_workgroups_are_bitwise_the_eight_wave_kernels(case):
    """pw6 / pw7 with 16-wave workgroups on the K >= 320 layers (round 4, option pw_waves16) against the 8-wave form (two M
    tiles per wave): forward (BN * SE + Swish prologue) and the data gradient of the transposed layer in all epilogue modes,
    y / dX and the statistics tiles BITWISE equal."""
    from x3dhip import _lib, ops
    dev = _dev()
    N, K, M, T, H, W = case
    x = _g(N, K, T, H, W, seed=1).float().to(dev)
    w = (_g(M, K, seed=2) / np.sqrt(K)).float().to(dev)              # forward: K -> M
    pre = torch.stack([1 + 0.2 * _g(N, K, seed=3), 0.3 * _g(N, K, seed=4)], -1).float().contiguous().to(dev)
    wp = ops.pw_pack(w)
    # data gradient of a conv M -> K (GEMM K = its Cout = K here, M rows = its Cin = M)
    wb = (_g(K, M, seed=5) / np.sqrt(K)).float().to(dev)
    wpt = ops.pw_pack(wb, transposed=True)
    g = _g(N, K, T, H, W, seed=6).float().to(dev)
    a = _g(N, K, T, H, W, seed=7).float().to(dev)
    cb = torch.stack([1 + 0.2 * _g(N, K, seed=8), 0.1 * _g(N, K, seed=9), 0.05 * _g(N, K, seed=10)], -1).float().contiguous().to(dev)
    xin = _g(N, M, T, H, W, seed=11).float().to(dev)
    prei = torch.stack([1 + 0.2 * _g(N, M, seed=12), 0.3 * _g(N, M, seed=13)], -1).float().contiguous().to(dev)
    res_out = torch.relu(_g(N, M, T, H, W, seed=14)).float().to(dev)
    res_raw = _g(N, M, T, H, W, seed=15).float().to(dev)
    add1 = _g(N, M, T, H, W, seed=16).float().to(dev)

    def run():
        outs = list(ops.pw_fwd(x, w, pre=pre, pre_act=2, wp=wp))
        outs.append(_lib.last_kernel())
        outs += list(ops.pw_fwd(x, w, wp=wp))
        outs.append(ops.pw_bwd_data(g, a, cb, wb, wpt=wpt)[0])
        outs += list(ops.pw_bwd_data(g, a, cb, wb, x=xin, pre=prei, pre_act=2, wpt=wpt))
        outs += list(ops.pw_bwd_data_res(g, a, cb, wb, res_out, res_raw, addend=add1, wpt=wpt))
        torch.cuda.synchronize()
        return outs

    with _lib.options(pw_waves16=0, no_pw8=1):
        r0 = run()
    for mode in (1, 2, 3):
        with _lib.options(pw_waves16=mode, no_pw8=1):
            r1 = run()
        assert r0[2] == "pw6_kernel" and r1[2] == "pw6_kernel"
        for i, (u, v) in enumerate(zip(r0, r1)):
            if i != 2:
                assert torch.equal(u, v), (mode, i)


def _terms(terms):
    """Backward GEMM operand split for the duration of a test: 3 bf16 terms (default, fp32 level) or 2 (~2^-16)."""
    from x3dhip import _lib
    return _lib.options(bwd_terms=terms)


# tolerance of the backward GEMM kernels against fp64: fp32-level with three terms, the split's 2^-16 with two
BTOL = {3: 2e-6, 2: TOL}


@pytest.mark.parametrize("terms", [3, 2])
@pytest.mark.parametrize("case", PW_CASES)
def test_pw_bwd(case, terms):
    with _terms(terms):
        _pw_bwd(case, BTOL[terms])


def _pw_bwd(case, btol):
    from x3dhip import ops
    dev = _dev()
    N, Ci, Co, T, H, W, s, act = case
    Ho, Wo = xo.out_hw(H, s), xo.out_hw(W, s)
    x = _g(N, Ci, T, H, W, seed=1)
    w = _g(Co, Ci, seed=2) / np.sqrt(Ci)
    pre = torch.stack([1 + 0.2 * _g(N, Ci, seed=3), 0.3 * _g(N, Ci, seed=4)], -1) if act else None
    g = _g(N, Co, T, Ho, Wo, seed=5)
    a = _g(N, Co, T, Ho, Wo, seed=6)
    cb = torch.stack([1 + 0.1 * _g(N, Co, seed=7), 0.1 * _g(N, Co, seed=8), 0.05 * _g(N, Co, seed=9)], -1)
    dY = cb[..., 0, None, None, None] * g + cb[..., 1, None, None, None] * a + cb[..., 2, None, None, None]
    sx = pre[..., 0, None, None, None] * x + pre[..., 1, None, None, None] if act else x
    xin = _act(sx, act).detach().requires_grad_(True)
    wv = w.view(Co, Ci, 1, 1, 1).clone().requires_grad_(True)
    (xo.pw(xin, wv, s) * dY).sum().backward()
    dw_ref, din_ref = wv.grad.view(Co, Ci), xin.grad
    to = lambda t: None if t is None else t.float().contiguous().to(dev)
    dw = ops.pw_bwd_weight(to(g), to(a), to(cb), to(x), (Co, Ci), stride=s, pre=to(pre), pre_act=act)
    assert _rel(dw, dw_ref) < btol
    if s == 1:
        addend = _g(N, Ci, T, H, W, seed=10)
        out_ref = (din_ref + addend) * (_dact(sx, act) if act else 1.0)
        wpt = ops.pw_pack(to(w), transposed=True)
        out, partial = ops.pw_bwd_data(to(g), to(a), to(cb), to(w), x=to(x) if act else None, pre=to(pre),
                                       pre_act=act, addend=to(addend))
        assert _rel(out, out_ref) < TOL
        out, partial = ops.pw_bwd_data(to(g), to(a), to(cb), to(w), x=to(x) if act else None, pre=to(pre),
                                       pre_act=act, addend=to(addend), wpt=wpt)
        assert _rel(out, out_ref) < (btol if act != 2 else TOL)      # (the Swish derivative uses the hardware exp / rcp)
        if act:
            st = partial.double().sum(2).cpu()
            assert _rel(st[..., 0], out_ref.sum(dim=(2, 3, 4))) < 1e-4
            assert _rel(st[..., 1], (out_ref * x).sum(dim=(2, 3, 4))) < 1e-4
        # stride-2 addend (the downsample branch's compact gradient)
        H2, W2 = xo.out_hw(H, 2), xo.out_hw(W, 2)
        add2 = _g(N, Ci, T, H2, W2, seed=11)
        full = torch.zeros(N, Ci, T, H, W, dtype=torch.float64)
        full[:, :, :, ::2, ::2] = add2
        out2, _ = ops.pw_bwd_data(to(g), to(a), to(cb), to(w), addend=to(add2), addend_stride=2, wpt=wpt)
        assert _rel(out2, din_ref + full) < btol
    else:
        # strided forward: its backward-data is computed densely at output resolution
        out, _ = ops.pw_bwd_data(to(g), to(a), to(cb), to(w), wpt=ops.pw_pack(to(w), transposed=True))
        assert _rel(out, F.conv_transpose3d(dY, w.view(Co, Ci, 1, 1, 1))) < btol


@pytest.mark.parametrize("case", [(2, 216, 96, 4, 10, 10), (2, 96, 216, 4, 14, 14), (2, 432, 192, 4, 5, 5), (8, 192, 432, 4, 7, 7),
                                  (4, 54, 24, 4, 28, 28), (4, 24, 54, 4, 28, 28), (2, 108, 48, 8, 14, 14)])
def test_three_term_backward_is_as_accurate_as_the_fp32_mfma_kernels(case):
    """"fp32 level" made concrete: on the same inputs, the error of the default (three bf16 terms, six products) backward
    kernels against the fp64 oracle is no larger than that of the exact fp32-MFMA kernels (options dgrad_f32 / wgrad_f32)
    against the same oracle -- both are fp32-accumulation noise (a few 1e-7); the two-term form is ~30x further away."""
    from x3dhip import _lib, ops
    dev = _dev()
    N, Ci, Co, T, H, W = case
    x = _g(N, Ci, T, H, W, seed=1)
    w = _g(Co, Ci, seed=2) / np.sqrt(Ci)
    g = _g(N, Co, T, H, W, seed=5)
    a = _g(N, Co, T, H, W, seed=6)
    cb = torch.stack([1 + 0.1 * _g(N, Co, seed=7), 0.1 * _g(N, Co, seed=8), 0.05 * _g(N, Co, seed=9)], -1)
    dY = cb[..., 0, None, None, None] * g + cb[..., 1, None, None, None] * a + cb[..., 2, None, None, None]
    din_ref = F.conv_transpose3d(dY, w.view(Co, Ci, 1, 1, 1))
    dw_ref = torch.einsum("nopqr,nipqr->oi", dY, x)
    to = lambda t: t.float().contiguous().to(dev)
    wpt = ops.pw_pack(to(w), transposed=True)

    def errs(**opts):
        with _lib.options(**opts):
            dx, _ = ops.pw_bwd_data(to(g), to(a), to(cb), to(w), wpt=wpt)
            dw = ops.pw_bwd_weight(to(g), to(a), to(cb), to(x), (Co, Ci))
            out = [_rel(dx, din_ref), _rel(dw, dw_ref)]
            if ops.pw_bwd_fused_ok(Ci, Co, T * H * W):
                dxf, _, dwf = ops.pw_bwd_fused(to(g), to(a), to(cb), (Co, Ci), wpt, to(x), mode=0)
                out += [_rel(dxf, din_ref), _rel(dwf, dw_ref)]
            return out
    e3, ef, e2 = errs(), errs(dgrad_f32=1, wgrad_f32=1), errs(bwd_terms=2)
    print("\n[%s] vs fp64: 3-term %s | fp32 MFMA %s | 2-term %s" % (case, ["%.1e" % v for v in e3], ["%.1e" % v for v in ef], ["%.1e" % v for v in e2]))
    for i in range(2):
        assert e3[i] <= 1.5 * ef[i] + 2e-7, (i, e3, ef)
    assert e2[1] > 5 * e3[1], (e2, e3)                 # (the option really changes the weight-gradient kernel ...
    if Co >= 64 and Ci >= 96:
        assert e2[0] > 5 * e3[0], (e2, e3)             #  ... and the data-gradient kernel of the large-channel layers)
    for v in e3:
        assert v < 1e-6


@pytest.mark.parametrize("terms", [3, 2])
@pytest.mark.parametrize("case", [c for c in PW_CASES if c[6] == 1])
def test_pw_bwd_data_res(case, terms):
    with _terms(terms):
        _pw_bwd_data_res(case, BTOL[terms])


def _pw_bwd_data_res(case, btol):
    """Data gradient with the producer block's residual-add + ReLU backward in the epilogue (every kernel variant the
    planner picks for these shapes, with and without packed weights, dense and stride-2 addend)."""
    from x3dhip import ops
    dev = _dev()
    N, Ci, Co, T, H, W, s, act = case
    w = _g(Co, Ci, seed=2) / np.sqrt(Ci)
    g = _g(N, Co, T, H, W, seed=5)
    a = _g(N, Co, T, H, W, seed=6)
    cb = torch.stack([1 + 0.1 * _g(N, Co, seed=7), 0.1 * _g(N, Co, seed=8), 0.05 * _g(N, Co, seed=9)], -1)
    dY = cb[..., 0, None, None, None] * g + cb[..., 1, None, None, None] * a + cb[..., 2, None, None, None]
    din = F.conv_transpose3d(dY, w.view(Co, Ci, 1, 1, 1))
    res_out = torch.relu(_g(N, Ci, T, H, W, seed=12))          # the producer block's output (about half zeros)
    res_raw = _g(N, Ci, T, H, W, seed=13)                       # its raw conv3 output
    addend = _g(N, Ci, T, H, W, seed=10)
    H2, W2 = xo.out_hw(H, 2), xo.out_hw(W, 2)
    add2 = _g(N, Ci, T, H2, W2, seed=11)
    full = torch.zeros(N, Ci, T, H, W, dtype=torch.float64)
    full[:, :, :, ::2, ::2] = add2
    to = lambda t: None if t is None else t.float().contiguous().to(dev)
    mask = (res_out > 0).double()
    for wpt in (None, ops.pw_pack(to(w), transposed=True)):
        for add, astride, ref in ((None, 1, din), (addend, 1, din + addend), (add2, 2, din + full)):
            ref = ref * mask
            out, partial = ops.pw_bwd_data_res(to(g), to(a), to(cb), to(w), to(res_out), to(res_raw), addend=to(add),
                                               addend_stride=astride, wpt=wpt)
            assert _rel(out, ref) < btol
            assert bool((out.cpu()[mask == 0] == 0).all())
            st = partial.double().sum(2).cpu()
            assert _rel(st[..., 0], ref.sum(dim=(2, 3, 4))) < 1e-4
            assert _rel(st[..., 1], (ref * res_raw).sum(dim=(2, 3, 4))) < 1e-4


FUSED_CASES = [
    # N, Cin, Cout, T, H, W, act of the conv's input prologue (mode 1) -- the stage 1-2 shapes of X3D-M plus tails
    (2, 24, 54, 4, 16, 16, 1),      # conv1 of stage 1 (64 x 32 tile)
    (2, 54, 24, 4, 16, 16, 2),      # conv3 of stage 1 (32 x 64 tile)
    (3, 48, 108, 2, 20, 20, 1),     # conv1 of stage 2 (128 x 64), P = 800: last chunk half empty
    (2, 108, 48, 2, 14, 14, 2),     # conv3 of stage 2 (64 x 128), P = 392: tail of 8 voxels
    (2, 24, 108, 4, 12, 12, 1),     # layer2.0 conv1 (128 x 32)
    (2, 40, 60, 2, 10, 10, 2),      # padded to (64, 64)
    (9, 24, 54, 2, 8, 8, 1),        # 18 chunks on a grid of 18
    (1, 60, 20, 1, 2, 2, 0),        # P = 4: a single partial chunk
    # N * ceil(P / 64) > 512: the PERSISTENT multi-chunk loop (a workgroup walks chunks c, c + 512, ...; one-chunk-ahead
    # register prefetch, LDS images re-staged per chunk) -- until round 3 only whole-network tests reached it.  These are the
    # stage 1-2 layers of train_M_2x4x158_s2 (GPUTEST_r02's red case):
    (2, 24, 54, 4, 79, 79, 1),      # layer1.0 conv1: P = 24 964 -> 782 chunks, last chunk 4 voxels, stride-2 addend on an odd plane
    (2, 54, 24, 4, 40, 40, 2),      # layer1.x conv3: P = 6400 -> 200 chunks
    (2, 24, 108, 4, 40, 40, 1),     # layer2.0 conv1 (stride-2 addend 40 -> 20)
    (2, 108, 48, 4, 20, 20, 2),     # layer2.x conv3
    (12, 48, 108, 4, 20, 20, 1),    # layer2.x conv1 at N = 12: 300 chunks; and
    (24, 54, 24, 4, 40, 40, 2),     # 2400 chunks on 512 workgroups: 4-5 chunks per workgroup, uneven tail round
    # stage 1-2 layers of the smallest golden clip (2, 4, 32, 32)
    (2, 24, 54, 4, 8, 8, 1), (2, 54, 24, 4, 8, 8, 2), (2, 48, 108, 4, 4, 4, 1), (2, 108, 48, 4, 4, 4, 2),
]


@pytest.mark.parametrize("terms", [3, 2])
@pytest.mark.parametrize("case", FUSED_CASES)
def test_pw_bwd_fused(case, terms):
    """x3d_pw_bwd_fused (data gradient + weight gradient in one pass, stages 1-2) against the fp64 evaluation of the
    same fused op, in its three epilogue modes, dense and stride-2 addend; three-term (default) and two-term operands."""
    with _terms(terms):
        _pw_bwd_fused(case, BTOL[terms])


@pytest.mark.parametrize("grid", [3, 40])
def test_pw_bwd_fused_long_chunk_loops(grid):
    """The persistent loop with MANY chunks per workgroup (option fb_grid: 3 or 40 workgroups instead of 512): 24-600 passes
    of the one-chunk-ahead prefetch / re-staged LDS images / statistics slots per workgroup, sample boundaries inside a
    workgroup's walk.  The forward stream kernel shares the option (contiguous chunk ranges, per-workgroup slots)."""
    from x3dhip import _lib, ops
    dev = _dev()
    with _lib.options(fb_grid=grid):
        for case in [(3, 24, 54, 4, 20, 20, 1), (3, 54, 24, 4, 20, 20, 2), (2, 48, 108, 2, 14, 14, 1)]:
            _pw_bwd_fused(case, BTOL[3])
        # forward stream kernel on the same grids
        N, Ci, Co, T, H, W = 3, 24, 54, 4, 20, 20
        x = _g(N, Ci, T, H, W, seed=1)
        w = _g(Co, Ci, seed=2) / np.sqrt(Ci)
        y_ref = xo.pw(x, w.view(Co, Ci, 1, 1, 1), 1)
        wd = w.float().to(dev)
        y, partial = ops.pw_fwd(x.float().to(dev), wd, wp=ops.pw_pack(wd))
        assert _rel(y, y_ref) < TOL
        st = partial.double().sum(2).cpu()
        assert _rel(st[..., 0], y_ref.sum(dim=(2, 3, 4))) < 5 * TOL + 1e-6
        assert _rel(st[..., 1], (y_ref ** 2).sum(dim=(2, 3, 4))) < 5 * TOL


def _pw_bwd_fused(case, btol):
    from x3dhip import ops
    dev = _dev()
    N, Ci, Co, T, H, W, act = case
    assert ops.pw_bwd_fused_ok(Ci, Co, T * H * W)
    x = _g(N, Ci, T, H, W, seed=1)
    w = _g(Co, Ci, seed=2) / np.sqrt(Ci)
    pre = torch.stack([1 + 0.2 * _g(N, Ci, seed=3), 0.3 * _g(N, Ci, seed=4)], -1)
    g = _g(N, Co, T, H, W, seed=5)
    a = _g(N, Co, T, H, W, seed=6)
    cb = torch.stack([1 + 0.1 * _g(N, Co, seed=7), 0.1 * _g(N, Co, seed=8), 0.05 * _g(N, Co, seed=9)], -1)
    dY = cb[..., 0, None, None, None] * g + cb[..., 1, None, None, None] * a + cb[..., 2, None, None, None]
    din = F.conv_transpose3d(dY, w.view(Co, Ci, 1, 1, 1))
    addend = _g(N, Ci, T, H, W, seed=10)
    H2, W2 = xo.out_hw(H, 2), xo.out_hw(W, 2)
    add2 = _g(N, Ci, T, H2, W2, seed=11)
    full = torch.zeros(N, Ci, T, H, W, dtype=torch.float64)
    full[:, :, :, ::2, ::2] = add2
    to = lambda t: None if t is None else t.float().contiguous().to(dev)
    wpt = ops.pw_pack(to(w), transposed=True)
    dwr = lambda xin: torch.einsum("nopqr,nipqr->oi", dY, xin)

    # mode 0: plain (+ addend); the conv's input is the materialised x
    P = T * H * W
    for add, astride, ref in ((None, 1, din), (addend, 1, din + addend), (add2, 2, din + full)):
        if not ops.pw_bwd_fused_ok(Ci, Co, P, 0, add is not None):     # (epilogue, addend) pairs whose LDS images do not fit
            continue                                                   # are refused (the engine then runs the separate kernels)
        dx, partial, dw = ops.pw_bwd_fused(to(g), to(a), to(cb), (Co, Ci), wpt, to(x), mode=0, addend=to(add),
                                           addend_stride=astride)
        assert partial is None
        assert _rel(dx, ref) < btol
        assert _rel(dw, dwr(x)) < btol
    # mode 1: activation backward of the conv's input (x raw, pre): conv3 (Swish), layer1.0 conv1 (ReLU of the stem)
    if act:
        sx = pre[..., 0, None, None, None] * x + pre[..., 1, None, None, None]
        for add, astride, base in ((None, 1, din), (add2, 2, din + full)):
            if not ops.pw_bwd_fused_ok(Ci, Co, P, 1, add is not None):
                continue
            ref = base * _dact(sx, act)
            dx, partial, dw = ops.pw_bwd_fused(to(g), to(a), to(cb), (Co, Ci), wpt, to(x), xpre=to(pre), xact=act, mode=1,
                                               addend=to(add), addend_stride=astride)
            assert _rel(dx, ref) < (btol if act != 2 else TOL)      # (Swish: hardware exp / rcp in the prologue and epilogue)
            assert _rel(dw, dwr(_act(sx, act))) < (btol if act != 2 else TOL)
            st = partial.double().sum(2).cpu()
            assert _rel(st[..., 0], ref.sum(dim=(2, 3, 4))) < 1e-4
            assert _rel(st[..., 1], (ref * x).sum(dim=(2, 3, 4))) < 1e-4
    # mode 2: residual-add + ReLU backward of the producing block (x = its output, ex = its raw conv3 output)
    xo_ = torch.relu(x)
    ex = _g(N, Ci, T, H, W, seed=13)
    mask = (xo_ > 0).double()
    if not ops.pw_bwd_fused_ok(Ci, Co, T * H * W, 2, True):       # residual mode: Cin <= 64 (two raw LDS tiles + the planes)
        return
    for add, astride, base in ((addend, 1, din + addend), (add2, 2, din + full)):
        ref = base * mask
        dx, partial, dw = ops.pw_bwd_fused(to(g), to(a), to(cb), (Co, Ci), wpt, to(xo_), mode=2, ex=to(ex), addend=to(add),
                                           addend_stride=astride)
        assert _rel(dx, ref) < btol
        assert bool((dx.cpu()[mask == 0] == 0).all())
        assert _rel(dw, dwr(xo_)) < btol
        st = partial.double().sum(2).cpu()
        assert _rel(st[..., 0], ref.sum(dim=(2, 3, 4))) < 1e-4
        assert _rel(st[..., 1], (ref * ex).sum(dim=(2, 3, 4))) < 1e-4


DW_CASES = [
    # N, C, T, H, W, stride
    (2, 6, 4, 14, 14, 1),
    (1, 5, 5, 13, 9, 1),      # W % 4 != 0
    (2, 4, 4, 16, 16, 2),
    (1, 3, 3, 15, 11, 2),     # odd sizes, stride 2
    (1, 7, 4, 7, 7, 1),       # several channels per block
    (1, 33, 2, 4, 4, 2),      # 16 channels per block + tail
    (1, 2, 3, 40, 56, 1),     # 3 row tiles
    (1, 2, 2, 58, 112, 2),    # wide stride-2 tile (layer1.0 geometry)
    (1, 2, 1, 9, 10, 1),      # T = 1
    (2, 9, 2, 24, 24, 2),     # the (16,2,47) multigrid-like chain: T = 2, small planes
    (2, 9, 2, 12, 12, 1),
    (2, 20, 2, 6, 6, 2),
    (2, 20, 2, 3, 3, 1),
    (2, 40, 2, 3, 3, 2),
    (2, 40, 2, 2, 2, 1),
    (1, 3, 2, 56, 56, 1),     # one channel per workgroup (scalar-weight variant), T = 2
    (1, 3, 4, 112, 112, 2),
    (16, 108, 2, 6, 6, 1),    # many small workgroups (caught a slot-reuse race in the T march)
    (2, 10, 16, 14, 14, 1),   # stage-3 geometry, T = 16
    (2, 3, 8, 28, 28, 1),     # stage-2 geometry
    (2, 3, 8, 28, 28, 2),     # stage-3.0 geometry (stride 2)
    (40, 20, 4, 7, 7, 1),     # many workgroups, tail group of channels
    (8, 108, 2, 28, 28, 1),   # base-shape stage-2 plane at full N, C (the tile-count queries take N, C: ABI 3)
    (8, 54, 2, 56, 56, 1),    # base-shape stage-1 plane
    (4, 216, 2, 40, 40, 1),   # 40 rows: balanced tiles of 14, 14, 12 rows
    # the channelwise layers of train_M_2x4x158_s2 (GPUTEST_r02's red case), N = 2, T = 4: odd 79-wide plane at stride 2 over
    # several row tiles (element-wise VW = 1 form), then the 40 -> 20 -> 10 -> 5 chain
    (2, 54, 4, 79, 79, 2),    # layer1.0: 79 -> 40, VW = 1, 7 row tiles backward
    (2, 54, 4, 40, 40, 1),    # layer1.x
    (2, 108, 4, 40, 40, 2),   # layer2.0: 40 -> 20
    (2, 108, 4, 20, 20, 1),
    (2, 216, 4, 20, 20, 2),   # layer3.0: 20 -> 10 (VW = 2)
    (2, 216, 4, 10, 10, 1),
    (2, 432, 4, 10, 10, 2),   # layer4.0: 10 -> 5 (VW = 1: odd output width)
    # the channelwise layers of the smallest golden clip (2, 4, 32, 32): planes 16 -> 8 -> 4 -> 2 -> 1
    (2, 54, 4, 16, 16, 2), (2, 54, 4, 8, 8, 1), (2, 108, 4, 8, 8, 2), (2, 108, 4, 4, 4, 1),
    (2, 216, 4, 4, 4, 2), (2, 216, 4, 2, 2, 1), (2, 432, 4, 2, 2, 2), (2, 432, 4, 1, 1, 1),
    (2, 432, 4, 5, 5, 1),
]


@pytest.mark.parametrize("case", DW_CASES)
def test_dw333_fwd_bwd(case):
    from x3dhip import ops
    dev = _dev()
    N, C, T, H, W, s = case
    Ho, Wo = xo.out_hw(H, s), xo.out_hw(W, s)
    x = _g(N, C, T, H, W, seed=1)
    w = _g(C, 1, 3, 3, 3, seed=2) / 3
    pre = torch.stack([1 + 0.2 * _g(N, C, seed=3), 0.3 * _g(N, C, seed=4)], -1)
    sx = pre[..., 0, None, None, None] * x + pre[..., 1, None, None, None]
    hin = torch.relu(sx).detach().requires_grad_(True)
    wv = w.clone().requires_grad_(True)
    y_ref = xo.dw333(hin, wv, s)
    to = lambda t: None if t is None else t.float().contiguous().to(dev)
    y, partial = ops.dw333_fwd(to(x), to(w), stride=s, pre=to(pre), pre_act=1)
    assert y.shape == y_ref.shape
    assert _rel(y, y_ref) < TOL
    st = partial.double().sum(2).cpu()
    assert _rel(st[..., 0], y_ref.sum(dim=(2, 3, 4))) < 1e-4
    assert _rel(st[..., 1], (y_ref ** 2).sum(dim=(2, 3, 4))) < 1e-4
    # backward
    g = _g(N, C, T, Ho, Wo, seed=5)
    a = y_ref.detach()
    cb = torch.stack([1 + 0.1 * _g(N, C, seed=7), 0.1 * _g(N, C, seed=8), 0.05 * _g(N, C, seed=9)], -1)
    dY = cb[..., 0, None, None, None] * g + cb[..., 1, None, None, None] * a + cb[..., 2, None, None, None]
    (y_ref * dY).sum().backward()
    out_ref = hin.grad * _dact(sx, 1)
    out, dw, bp = ops.dw333_bwd(to(g), to(a), to(cb), to(w), to(x), stride=s, pre=to(pre), pre_act=1)
    assert _rel(out, out_ref) < TOL
    assert _rel(dw, wv.grad) < TOL
    st = bp.double().sum(2).cpu()
    assert _rel(st[..., 0], out_ref.sum(dim=(2, 3, 4))) < 1e-4
    assert _rel(st[..., 1], (out_ref * x).sum(dim=(2, 3, 4))) < 1e-4


@pytest.mark.parametrize("case", [(8, 432, 16, 7, 7, 1), (2, 10, 16, 14, 14, 1), (2, 3, 8, 28, 28, 2), (1, 7, 9, 7, 7, 1),
                                  (2, 40, 11, 5, 5, 2), (1, 2, 8, 40, 56, 1), (2, 5, 18, 14, 14, 2), (1, 3, 17, 7, 7, 1)])
def test_dw333_t_segments_equal_the_single_march(case):
    """Launches with fewer workgroups than CUs cut the T march into two segments (options dw_tsplit_wgs / dw_tsplit_wgs_fwd: the 7 x 7
    planes of stage 4 at the base shape in both directions, the 14 x 14 planes of stage 3 forward).  Every output voxel is computed by the same arithmetic either way: y and dx BITWISE equal to
    the unsplit launch (thresholds 0); statistics and weight gradients are sums over more partials: 1e-6."""
    from x3dhip import _lib, ops
    dev = _dev()
    N, C, T, H, W, s = case
    Ho, Wo = xo.out_hw(H, s), xo.out_hw(W, s)
    to = lambda t: t.float().contiguous().to(dev)
    x, w = to(_g(N, C, T, H, W, seed=1)), to(_g(C, 1, 3, 3, 3, seed=2) / 3)
    pre = to(torch.stack([1 + 0.2 * _g(N, C, seed=3), 0.3 * _g(N, C, seed=4)], -1))
    g, a = to(_g(N, C, T, Ho, Wo, seed=5)), to(_g(N, C, T, Ho, Wo, seed=6))
    cb = to(torch.stack([1 + 0.1 * _g(N, C, seed=7), 0.1 * _g(N, C, seed=8), 0.05 * _g(N, C, seed=9)], -1))

    def run(wgs, quad=0):
        with _lib.options(dw_tsplit_wgs=wgs, dw_tsplit_wgs_fwd=wgs, dw_tquad_wgs=quad, dw_tquad_wgs_fwd=quad):
            y, p = ops.dw333_fwd(x, w, stride=s, pre=pre, pre_act=1)
            dx, dw, bp = ops.dw333_bwd(g, a, cb, w, x, stride=s, pre=pre, pre_act=1)
            return y, p, dx, dw, bp
    y0, p0, dx0, dw0, bp0 = run(0)
    y1, p1, dx1, dw1, bp1 = run(1 << 20)                     # always split (T >= 8)
    assert p1.shape[2] == 2 * p0.shape[2] and bp1.shape[2] == 2 * bp0.shape[2]
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    assert _rel(p1.double().sum(2), p0.double().sum(2)) < 1e-6 and _rel(bp1.double().sum(2), bp0.double().sum(2)) < 1e-6
    assert _rel(dw1, dw0) < 1e-6
    if T >= 16:                                              # round 4: four segments (options dw_tquad_wgs[_fwd])
        y2, p2, dx2, dw2, bp2 = run(1 << 20, 1 << 20)
        assert p2.shape[2] == 4 * p0.shape[2] and bp2.shape[2] == 4 * bp0.shape[2]
        assert torch.equal(y0, y2) and torch.equal(dx0, dx2)
        assert _rel(p2.double().sum(2), p0.double().sum(2)) < 1e-6 and _rel(bp2.double().sum(2), bp0.double().sum(2)) < 1e-6
        assert _rel(dw2, dw0) < 1e-6


@pytest.mark.parametrize("case", DW_CASES)
@pytest.mark.parametrize("S", [1, 2])
def test_dw333_fwd_stats_equals_finalize_then_conv(case, S):
    """The training form with bn1's finalize folded into the depthwise prologue equals bn_fwd_finalize + dw333_fwd:
    coefficients / saved statistics / running statistics to fp64-reorder noise, the conv output to fp32 rounding."""
    from x3dhip import ops
    dev = _dev()
    N, C, T, H, W, s = case
    if N % S:
        pytest.skip("batch not divisible by the split count")
    x = _g(N, C, T, H, W, seed=1).float().to(dev)
    w = (_g(C, 1, 3, 3, 3, seed=2) / 3).float().to(dev)
    P = T * H * W
    for stiles in (1, 7, 200):
        sp = torch.stack([_g(N, C, stiles, seed=11) * 3, 50 + _g(N, C, stiles, seed=12).abs() * 40], -1).float().to(dev)
        gamma = (1 + 0.2 * _g(C, seed=13)).float().to(dev)
        beta = (0.3 * _g(C, seed=14)).float().to(dev)
        rm0 = (0.1 * _g(S, C, seed=15)).float().to(dev)
        rv0 = (1 + 0.1 * _g(S, C, seed=16).abs()).float().to(dev)
        rm_a, rv_a, rm_b, rv_b = rm0.clone(), rv0.clone(), rm0.clone(), rv0.clone()
        coef, save, _ = ops.bn_fwd_finalize(sp, S, P, gamma, beta, rm_a, rv_a, 0.1, 1e-5)
        y_ref, p_ref = ops.dw333_fwd(x, w, stride=s, pre=coef, pre_act=1)
        y, part, coef2, save2 = ops.dw333_fwd_stats(x, w, sp, S, P, gamma, beta, rm_b, rv_b, stride=s, pre_act=1,
                                                    momentum=0.1, eps=1e-5)
        assert _rel(coef2, coef) < 1e-6
        assert _rel(save2, save) < 1e-6
        assert _rel(rm_b, rm_a) < 1e-6 and _rel(rv_b, rv_a) < 1e-6
        assert _rel(y, y_ref) < 1e-5
        assert _rel(part.double().sum(2), p_ref.double().sum(2)) < 1e-5


# (N, C, Wd, tiles, S): the SE blocks of X3D-M (54 / 108 / 216 / 432 channels, widths 8 / 8 / 16 / 32), the large-batch
# multigrid shapes (N = 64 / 128 with 8 splits), XL widths (162 -> 16, 630 -> 40: Wd % 4 == 0 and a 3-unit-per-wave form),
# an odd width (scalar fc2 loads), C = 1024
SE_BN_CASES = [(8, 54, 8, 4, 1), (8, 108, 8, 2, 1), (8, 216, 16, 2, 1), (8, 432, 32, 2, 1), (2, 432, 32, 1, 2),
               (64, 54, 8, 1, 8), (128, 216, 16, 1, 8), (16, 108, 8, 3, 2), (4, 162, 16, 2, 1), (4, 630, 40, 1, 2),
               (2, 306, 20, 5, 1), (3, 70, 7, 2, 3), (2, 1024, 64, 1, 1)]


@pytest.mark.parametrize("case", SE_BN_CASES)
def test_elementwise_se_bn_fwd_equals_finalize_then_se(case):
    """x3d_se_bn_fwd (round 4: bn2's finalize + the SE branch in one launch) against the two launches it replaces
    (x3d_bn_fwd_finalize + x3d_se_fwd) and against an fp64 evaluation of x3d.py:47-58,153-159 on the same partial sums."""
    from x3dhip import ops
    dev = _dev()
    N, C, Wd, tiles, S = case
    count = 3136
    sp = torch.stack([_g(N, C, tiles, seed=21) * 30, 800 + _g(N, C, tiles, seed=22).abs() * 400], -1).float().to(dev)
    gamma = (1 + 0.2 * _g(C, seed=23)).float().to(dev)
    beta = (0.3 * _g(C, seed=24)).float().to(dev)
    w1 = (_g(Wd, C, seed=25) / C ** 0.5).float().to(dev)
    b1 = (0.1 * _g(Wd, seed=26)).float().to(dev)
    w2 = (_g(C, Wd, seed=27) / Wd ** 0.5).float().to(dev)
    b2 = (0.1 * _g(C, seed=28)).float().to(dev)
    rm0 = (0.1 * _g(S, C, seed=29)).float().to(dev)
    rv0 = (1 + 0.1 * _g(S, C, seed=30).abs()).float().to(dev)
    rm_a, rv_a, rm_b, rv_b = rm0.clone(), rv0.clone(), rm0.clone(), rv0.clone()
    coef, save_a, nsum_a = ops.bn_fwd_finalize(sp, S, count, gamma, beta, rm_a, rv_a, 0.1, 1e-5, want_nsum=True)
    ce_a, se_a, z_a, pool_a = ops.se_fwd(coef, nsum_a, count, w1, b1, w2, b2)
    ce_b, save_b, nsum_b, se_b, z_b, pool_b = ops.se_bn_fwd(sp, S, count, gamma, beta, rm_b, rv_b, w1, b1, w2, b2, 0.1, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(save_b, save_a) and torch.equal(nsum_b, nsum_a)          # fp64 sums of fp32 terms: exact either way
    assert torch.equal(rm_b, rm_a) and torch.equal(rv_b, rv_a)
    assert torch.equal(pool_b, pool_a)
    for a, b in ((z_a, z_b), (se_a, se_b), (ce_a, ce_b)):                        # fc1's summation order differs
        assert _rel(b, a) < 2e-6
    # fp64 truth
    d = sp.double().cpu().sum(2)                                                  # [N, C, 2]
    g64, b64 = gamma.double().cpu(), beta.double().cpu()
    ref_ce = torch.empty(N, C, 2, dtype=torch.float64)
    ref_se = torch.empty(N, C, dtype=torch.float64)
    for j in range(S):
        idx = torch.arange(j, N, S)
        cnt = count * len(idx)
        mean = d[idx, :, 0].sum(0) / cnt
        var = (d[idx, :, 1].sum(0) / cnt - mean * mean).clamp_min(0)
        invstd = 1 / torch.sqrt(var + 1e-5)
        sc, sh = g64 * invstd, b64 - mean * g64 * invstd
        pool = sc * d[idx, :, 0] / count + sh                                     # [ns, C]
        z = torch.relu(pool @ w1.double().cpu().t() + b1.double().cpu())
        se = torch.sigmoid(z @ w2.double().cpu().t() + b2.double().cpu())
        ref_se[idx] = se
        ref_ce[idx, :, 0] = sc * se
        ref_ce[idx, :, 1] = sh * se
        assert _rel(save_b[0, j], mean) < 1e-6 and _rel(save_b[1, j], invstd) < 1e-6
    assert _rel(se_b, ref_se) < 2e-6
    assert _rel(ce_b, ref_ce) < 2e-6


# (N, C, Wd, tiles, S): SE blocks of stages 3-4 at the multigrid shapes (98 / 25 / 13 / 7 statistics tiles per row), XL widths
SE_BWD_CASES = [(8, 216, 16, 98, 1), (8, 432, 32, 25, 1), (16, 216, 16, 49, 2), (64, 432, 32, 7, 8), (2, 216, 16, 13, 2),
                (4, 306, 20, 50, 1), (2, 630, 40, 25, 1), (3, 70, 7, 100, 3), (2, 1024, 64, 30, 1)]


@pytest.mark.parametrize("case", SE_BWD_CASES)
def test_elementwise_se_bwd_merged_sample_kernel_equals_the_two_launches(case):
    """x3d_se_bn_bwd_finalize with the merged tile-reduction + per-sample SE backward kernel (round 4) against the same entry
    point with option no_se_bwd_merge (reduce_tiles + se_bwd_sample launches): BN-backward coefficients, dgamma / dbeta and
    the SE weight gradients agree to fp32 summation-order noise."""
    from x3dhip import _lib, ops
    dev = _dev()
    N, C, Wd, tiles, S = case
    count = 32 * tiles
    part = torch.stack([_g(N, C, tiles, seed=31) * 0.3, _g(N, C, tiles, seed=32) * 2], -1).float().to(dev)
    gamma = (1 + 0.2 * _g(C, seed=33)).float().to(dev)
    beta = (0.3 * _g(C, seed=34)).float().to(dev)
    save = torch.stack([0.2 * _g(S, C, seed=35), 0.5 + _g(S, C, seed=36).abs()], 0).float().contiguous().to(dev)
    nsum = (_g(N, C, seed=37) * count * 0.3).float().to(dev)
    w1 = (_g(Wd, C, seed=38) / C ** 0.5).float().to(dev)
    w2 = (_g(C, Wd, seed=39) / Wd ** 0.5).float().to(dev)
    se = torch.sigmoid(_g(N, C, seed=40)).float().to(dev)
    z = torch.relu(_g(N, Wd, seed=41)).float().to(dev)
    pool = _g(N, C, seed=42).float().to(dev)
    with _lib.options(no_se_bwd_merge=1):
        cb0, o0 = ops.se_bn_bwd_finalize(part, S, count, gamma, beta, save, nsum, w1, w2, se, z, pool)
    cb1, o1 = ops.se_bn_bwd_finalize(part, S, count, gamma, beta, save, nsum, w1, w2, se, z, pool)
    torch.cuda.synchronize()
    assert _rel(cb1, cb0) < 2e-6
    for k in o0:
        assert _rel(o1[k], o0[k]) < 5e-6, k


@pytest.mark.parametrize("shape", [(2, 3, 4, 16, 16), (1, 3, 3, 15, 11), (1, 3, 2, 64, 64), (2, 3, 4, 158, 158)])
def test_stem(shape):
    from x3dhip import ops
    dev = _dev()
    N, Ci, T, H, W = shape
    Co = 24
    x = _g(N, Ci, T, H, W, seed=1)
    ws = (_g(Co, Ci, 1, 3, 3, seed=2) / 5).requires_grad_(True)
    wt = (_g(Co, 1, 5, 1, 1, seed=3) / 2).requires_grad_(True)
    ys = xo.stem133(x, ws)
    ys_leaf = ys.detach().requires_grad_(True)
    yt = xo.dw5t(ys_leaf, wt)
    to = lambda t: t.detach().float().contiguous().to(dev)
    ys_h = ops.stem133_fwd(to(x), to(ws))
    assert _rel(ys_h, ys) < TOL
    yt_h, partial = ops.dw5t_fwd(ys_h, to(wt))
    assert _rel(yt_h, yt) < TOL
    st = partial.double().sum(2).cpu()
    assert _rel(st[..., 0], yt.detach().sum(dim=(2, 3, 4))) < 1e-4
    assert _rel(st[..., 1], (yt.detach() ** 2).sum(dim=(2, 3, 4))) < 1e-4
    g = _g(*yt.shape, seed=5)
    cb = torch.stack([1 + 0.1 * _g(N, Co, seed=7), 0.1 * _g(N, Co, seed=8), 0.05 * _g(N, Co, seed=9)], -1)
    dY = cb[..., 0, None, None, None] * g + cb[..., 1, None, None, None] * yt.detach() + cb[..., 2, None, None, None]
    (yt * dY).sum().backward()
    dx_h, dwt_h = ops.dw5t_bwd(to(g), to(yt), to(cb), to(wt), to(ys))
    assert _rel(dx_h, ys_leaf.grad) < TOL
    assert _rel(dwt_h, wt.grad) < TOL
    (ys * ys_leaf.grad).sum().backward()
    dws_h = ops.stem133_bwd_weight(to(x), dx_h, ws.shape)
    assert _rel(dws_h, ws.grad) < TOL


def test_elementwise_and_sgd():
    from x3dhip import ops
    dev = _dev()
    for (N, C, P) in [(2, 5, 4 * 7 * 7), (1, 3, 3 * 5 * 5), (2, 4, 5000)]:
        a3 = _g(N, C, P, 1, 1, seed=1)
        res = _g(N, C, P, 1, 1, seed=2)
        c3 = torch.stack([1 + 0.2 * _g(N, C, seed=3), 0.3 * _g(N, C, seed=4)], -1)
        cd = torch.stack([1 + 0.2 * _g(N, C, seed=5), 0.3 * _g(N, C, seed=6)], -1)
        to = lambda t: None if t is None else t.float().contiguous().to(dev)
        bc = lambda c, k: c[..., k, None, None, None]
        for use_cd in (False, True):
            ref = torch.relu(bc(c3, 0) * a3 + bc(c3, 1) + ((bc(cd, 0) * res + bc(cd, 1)) if use_cd else res))
            out = ops.bn_add_relu_fwd(to(a3), to(c3), to(res), to(cd) if use_cd else None)
            assert _rel(out, ref) < 1e-6
            dout = _g(N, C, P, 1, 1, seed=7)
            g_ref = dout * (ref > 0)
            g, p1, p2 = ops.bn_add_relu_bwd(to(dout), out, to(a3), to(res) if use_cd else None)
            assert _rel(g, g_ref) < 1e-6
            st = p1.double().sum(2).cpu()
            assert _rel(st[..., 0], g_ref.sum(dim=(2, 3, 4))) < 1e-4
            assert _rel(st[..., 1], (g_ref * a3).sum(dim=(2, 3, 4))) < 1e-4
            if use_cd:
                st2 = p2.double().sum(2).cpu()
                assert _rel(st2[..., 1], (g_ref * res).sum(dim=(2, 3, 4))) < 1e-4
        pooled = ops.bn_relu_pool_fwd(to(a3), to(c3))
        pref = torch.relu(bc(c3, 0) * a3 + bc(c3, 1)).mean(dim=(2, 3, 4))
        assert _rel(pooled, pref) < 1e-5
        dp = _g(N, C, seed=8)
        gp, pp = ops.bn_relu_pool_bwd(to(a3), to(c3), to(dp))
        gref = dp[..., None, None, None] / P * ((bc(c3, 0) * a3 + bc(c3, 1)) > 0)
        assert _rel(gp, gref) < 1e-6
        assert _rel(pp.double().sum(2).cpu()[..., 1], (gref * a3).sum(dim=(2, 3, 4))) < 1e-4
    # fused SGD vs torch.optim.SGD semantics
    w = _g(1000, seed=1).float()
    gr = _g(1000, seed=2).float()
    p = torch.nn.Parameter(w.clone())
    opt = torch.optim.SGD([p], lr=0.1, momentum=0.9, weight_decay=5e-5)
    wd, md = w.clone().to(dev), torch.zeros(1000, device=dev)
    for it in range(3):
        p.grad = gr.clone() * (it + 1)
        opt.step()
        ops.sgd_fused(wd, (gr * (it + 1)).to(dev), md, 0.1, first=(it == 0))
    assert _rel(wd, p.detach()) < 1e-6


def test_reduce_partials_batch_is_bitwise_the_single_reductions():
    """One launch for many (partial, out) jobs (more than one kernel-argument batch of 96) == per-job reductions."""
    from x3dhip import ops
    dev = _dev()
    jobs, refs = [], []
    for j in range(130):
        groups, n = 1 + (j * 7) % 45, 1 + (j * 131) % 700
        p = _g(groups, n, seed=100 + j).float().to(dev)
        o = torch.full((n,), float("nan"), device=dev)
        jobs.append((p, o))
        refs.append(ops.reduce_partials(p, n))
    ops.reduce_partials_batch(jobs)
    for (p, o), r in zip(jobs, refs):
        assert torch.equal(o, r)
        assert _rel(o, p.double().sum(0)) < 1e-6


def test_pw_bwd_weight_batch_is_bitwise_the_single_launches():
    """Every PW case (all tile variants, strided gathers, the non-tiled fallback shapes) twice over -- more than one
    24-job kernel-argument batch per variant -- postponed into DeferredGrads and flushed: bitwise the per-conv results."""
    from x3dhip import ops
    dev = _dev()
    to = lambda t: None if t is None else t.float().contiguous().to(dev)
    d = ops.DeferredGrads()
    outs, refs = [], []
    for rep in range(4):
        for ci, case in enumerate(PW_CASES):
            N, Ci, Co, T, H, W, s, act = case
            if N > 3 and rep > 0:
                continue
            Ho, Wo = xo.out_hw(H, s), xo.out_hw(W, s)
            sd = 1000 * rep + 10 * ci
            x = to(_g(N, Ci, T, H, W, seed=sd + 1))
            pre = to(torch.stack([1 + 0.2 * _g(N, Ci, seed=sd + 3), 0.3 * _g(N, Ci, seed=sd + 4)], -1)) if act else None
            g, a = to(_g(N, Co, T, Ho, Wo, seed=sd + 5)), to(_g(N, Co, T, Ho, Wo, seed=sd + 6))
            cb = to(torch.stack([1 + 0.1 * _g(N, Co, seed=sd + 7), 0.1 * _g(N, Co, seed=sd + 8), 0.05 * _g(N, Co, seed=sd + 9)], -1))
            refs.append(ops.pw_bwd_weight(g, a, cb, x, (Co, Ci), stride=s, pre=pre, pre_act=act))
            outs.append(ops.pw_bwd_weight(g, a, cb, x, (Co, Ci), stride=s, pre=pre, pre_act=act, defer=d))
    assert len(d.wjobs) == len(outs) > 48
    d.flush()
    assert not d.wjobs and not d.reduces
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)


@pytest.mark.parametrize("case", DW_CASES)
def test_dw333_bwd_stats_equals_finalize_then_conv(case):
    """BN-backward finalize (single split) folded into the depthwise backward prologue == bn_bwd_finalize + dw333_bwd."""
    from x3dhip import ops
    dev = _dev()
    N, C, T, H, W, s = case
    Ho, Wo = xo.out_hw(H, s), xo.out_hw(W, s)
    to = lambda t: t.float().contiguous().to(dev)
    x = to(_g(N, C, T, H, W, seed=1))
    w = to(_g(C, 1, 3, 3, 3, seed=2) / 3)
    pre = to(torch.stack([1 + 0.2 * _g(N, C, seed=3), 0.3 * _g(N, C, seed=4)], -1))
    g, a = to(_g(N, C, T, Ho, Wo, seed=5)), to(_g(N, C, T, Ho, Wo, seed=6))
    gamma = to(1 + 0.2 * _g(C, seed=13))
    save = to(torch.stack([0.1 * _g(1, C, seed=15), 1 + 0.1 * _g(1, C, seed=16).abs()], 0))      # [2, 1, C]
    P = T * Ho * Wo
    for stiles in (1, 5, 130):
        sp = to(torch.stack([_g(N, C, stiles, seed=11), 2 * _g(N, C, stiles, seed=12)], -1))
        cb, dg_ref, db_ref = ops.bn_bwd_finalize(sp, 1, P, gamma, save)
        out_ref, dw_ref, bp_ref = ops.dw333_bwd(g, a, cb, w, x, stride=s, pre=pre, pre_act=1)
        dg, db = torch.full_like(gamma, float("nan")), torch.full_like(gamma, float("nan"))
        out, dw, bp = ops.dw333_bwd(g, a, None, w, x, stride=s, pre=pre, pre_act=1, bn=(sp, P, gamma, save, dg, db))
        assert _rel(dg, dg_ref) < 1e-6 and _rel(db, db_ref) < 1e-6
        assert _rel(out, out_ref) < 1e-5
        assert _rel(dw, dw_ref) < 1e-5
        assert _rel(bp.double().sum(2), bp_ref.double().sum(2)) < 1e-5


@pytest.mark.parametrize("R,K,J,C", [(8, 432, 2048, 400), (3, 630, 2048, 157), (20, 432, 2048, 400), (70, 48, 64, 10)])
def test_head_kernels(R, K, J, C):
    """csrc/head.hip (fc1 -> ReLU -> Dropout -> fc2, mean cross entropy, and their backward: x3d.py:333-339,
    train_x3d_kinetics_multigrid.py:189,259) against an fp64 evaluation; with dropout the kernel's own mask (recovered from
    its output) is fed to the reference, and the mask statistics / per-step redraw are checked."""
    from x3dhip import ops
    dev = _dev()
    pooled = torch.relu(_g(R, K, seed=1))
    w1 = _g(J, K, seed=2) / np.sqrt(K)
    w2 = _g(C, J, seed=3) / np.sqrt(J)
    b2 = 0.1 * _g(C, seed=4)
    labels = torch.randint(0, C, (R,), generator=torch.Generator().manual_seed(5))
    to = lambda t: t.float().contiguous().to(dev)
    for p in (0.0, 0.5):
        rng = ops.head_rng_state(dev, seed=1234) if p > 0 else None
        hd, logits = ops.head_fwd(to(pooled), to(w1), to(w2), to(b2), p, rng)
        h_ref = torch.relu(pooled @ w1.t())
        if p > 0:
            mask = (hd.cpu().double() != 0) | (h_ref <= 1e-9)              # kept elements (zeros of the ReLU are undetermined)
            frac = float(((hd.cpu() != 0) & (h_ref > 1e-3)).sum()) / float((h_ref > 1e-3).sum())
            assert abs(frac - 0.5) < 0.03, frac
            hd_ref = h_ref * mask.double() / (1 - p)
        else:
            hd_ref = h_ref
        assert _rel(hd, hd_ref) < TOL
        lg_ref = (hd_ref @ w2.t() + b2).detach().requires_grad_(True)
        assert _rel(logits, lg_ref) < TOL
        loss_ref = F.cross_entropy(lg_ref, labels)
        loss_ref.backward()
        loss, dlog = ops.head_ce(logits, labels.to(dev), rng)
        assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
        assert _rel(dlog, lg_ref.grad) < 1e-4
        dlg = lg_ref.grad
        dhd = dlg @ w2
        dh = dhd * (hd_ref > 0).double() / ((1 - p) if p > 0 else 1.0)
        dpooled, dw1, dw2, db2 = ops.head_bwd(to(dlg), hd, to(pooled), to(w1), to(w2), p)
        assert _rel(dw2, dlg.t() @ hd_ref) < TOL
        assert _rel(db2, dlg.sum(0)) < TOL
        assert _rel(dw1, dh.t() @ pooled) < TOL
        assert _rel(dpooled, dh @ w1) < TOL
        if p > 0:
            assert int(rng[1]) == 1                                        # head_ce advanced the draw counter
            hd2, _ = ops.head_fwd(to(pooled), to(w1), to(w2), to(b2), p, rng)
            assert float(((hd2 != 0) != (hd != 0)).float().mean()) > 0.2   # a fresh mask
