"""CPU execution of a lowered hsefr plan with the ORACLE's NumPy ops -- test infrastructure.

Lets the CPU suite check the product's graph lowering (BN folding, weight decode, padding,
buffer assignment, serialisation) against the unfused graph interpreter without a GPU.  It
reads the plan back from its SERIALISED bytes, so the struct layout and the weight layouts the
kernels expect (pointwise kernel transposed, depthwise squeezed) are what is being executed.
"""
import struct

import numpy as np

from oracle import tf_graph as tfo

HEADER = struct.Struct("<QIIIIII3i3IQ")
BUFFER = struct.Struct("<QII")
OP = struct.Struct("<II3i3i3i3i2iii5Q")
NO_OFFSET = 0xFFFFFFFFFFFFFFFF


def _act(x, act):
    if act == 1:
        return np.maximum(x, 0)
    if act == 2:
        return np.minimum(np.maximum(x, 0), 6)
    if act == 3:
        return 1.0 / (1.0 + np.exp(-x))
    return x


def parse(blob: bytes):
    h = HEADER.unpack_from(blob, 0)
    magic, version, n_buf, n_ops, in_h, in_w, in_c = h[:7]
    out_buf, out_elems, blob_bytes = h[7:10], h[10:13], h[13]
    off = HEADER.size
    bufs = [BUFFER.unpack_from(blob, off + i * BUFFER.size) for i in range(n_buf)]
    off += n_buf * BUFFER.size
    ops = [OP.unpack_from(blob, off + i * OP.size) for i in range(n_ops)]
    off += n_ops * OP.size
    data = blob[off:]
    assert len(data) == blob_bytes
    return dict(magic=magic, version=version, in_hwc=(in_h, in_w, in_c), out_buf=out_buf, out_elems=out_elems,
                bufs=bufs, ops=ops, data=data)


def PW(a, wt):
    """pointwise product hook: a [M,K] . wt[N,K]^T (tests of split-precision arithmetic replace it)"""
    return a.dot(wt.T)


def run(blob: bytes, x: np.ndarray, dtype=np.float64, check_buffers=True):
    p = parse(blob)
    n = x.shape[0]
    data = p["data"]

    def arr(off, count):
        assert off != NO_OFFSET and off % 16 == 0
        return np.frombuffer(data, np.float32, count, off).astype(dtype)

    def arr_bf16(off, count):
        assert off != NO_OFFSET and off % 16 == 0
        u = np.frombuffer(data, np.uint16, count, off).astype(np.uint32) << 16
        return u.view(np.float32).astype(dtype)

    from oracle.resnet50 import bf16_round

    mem = {}
    owner = {}
    split_fmt = {}          # buffer -> a_log2 of the split rows it holds (0 = plain fp32): the storage-format contract
    x = x.astype(dtype)
    for i, o in enumerate(p["ops"]):
        (kind, act, in_buf, out_buf, res_buf, h, w, cin, oh, ow, cout, kh, kw, stride, pad_t, pad_l, _r, flags,
         w_off, sc_off, sh_off, w2_off, sh2_off) = o
        src = x if in_buf == -1 else mem[in_buf]
        if i > 0 and p["ops"][i - 1][17] & 4 and in_buf == p["ops"][i - 1][3]:
            src = in_registers      # the op behind an OUT_SUB2 pair reads the full map (from registers), not the stored compact one
        src = src.reshape(n, h, w, cin)
        sub2 = bool(flags & 4)
        res = None if res_buf < 0 else mem[res_buf]
        assert in_buf != out_buf, "op %d writes the buffer it reads" % i
        assert res_buf != out_buf
        if check_buffers:
            assert p["bufs"][out_buf][0] * p["bufs"][out_buf][1] >= oh * ow * cout * (2 if kind in (7, 8, 10, 20) else 4)
        if sub2:                    # hsefr_op_flags OUT_SUB2: computed at every pixel, stored at even rows / columns
            assert flags & 1 and kind == 7 and kh == 1 and kw == 1 and (oh, ow) == ((h + 1) // 2, (w + 1) // 2)
            oh, ow = h, w
        if kind == 1:
            k = arr(w_off, kh * kw * cin * cout).reshape(kh, kw, cin, cout)
            pb = (oh - 1) * stride + kh - h - pad_t
            pr = (ow - 1) * stride + kw - w - pad_l
            y = tfo.conv2d(src, k, (stride, stride), "", explicit_pads=(pad_t, max(pb, 0), pad_l, max(pr, 0)))
            y = _act(y + arr(sh_off, cout), act)
        elif kind == 2:
            k = arr(w_off, 9 * cin).reshape(3, 3, cin, 1)
            pb = max((oh - 1) * stride + 3 - h - pad_t, 0)
            pr = max((ow - 1) * stride + 3 - w - pad_l, 0)
            xp = np.pad(src, ((0, 0), (pad_t, pb), (pad_l, pr), (0, 0)))
            y = tfo.depthwise_conv2d(xp, k, (stride, stride), "VALID")
            y = _act(y * arr(sc_off, cin) + arr(sh_off, cin), act)
            if _r:      # result stored as split rows for the GEMM behind it: same values, another storage format
                assert act == 2 and cin % 32 == 0 and 0 < _r <= 12, "split-row depthwise output needs ReLU6, c % 32 == 0"
        elif kind == 3:
            wt = arr(w_off, cin * cout).reshape(cout, cin)
            y = _act(PW(src.reshape(-1, cin), wt) + arr(sh_off, cout), act).reshape(n, oh, ow, cout)
        elif kind in (12, 16):     # split-f16 pointwise (16: on pre-split input): the fp64 weights the two f16 planes stand for
            assert split_fmt.get(in_buf, 0) == (_r if kind == 16 else 0), "op %d reads buffer %d in the wrong storage format" % (i, in_buf)
            assert kind == 12 or (cin % 32 == 0 and cout % 128 == 0)
            from hse_facerec_tf_amd.lowering import unsplit_pointwise_weights
            img = np.frombuffer(data, np.uint16, cout * cin * 2, w_off).reshape(cout, cin // 32, 64)
            wt = unsplit_pointwise_weights(img, np.frombuffer(data, np.float32, cout, sc_off), _r).astype(dtype)
            assert 0 < _r <= 24 and float(np.abs(src).max()) * 2.0 ** _r < 32768, "split-f16 input bound violated"
            y = _act(PW(src.reshape(-1, cin), wt) + arr(sh_off, cout), act).reshape(n, oh, ow, cout)
        elif kind == 22:     # pre-split pointwise + global average pool
            assert split_fmt.get(in_buf, 0) == _r and cin % 32 == 0 and cout % 128 == 0 and 33 <= h * w <= 288 and (oh, ow) == (1, 1)
            from hse_facerec_tf_amd.lowering import unsplit_pointwise_weights
            img = np.frombuffer(data, np.uint16, cout * cin * 2, w_off).reshape(cout, cin // 32, 64)
            wt = unsplit_pointwise_weights(img, np.frombuffer(data, np.float32, cout, sc_off), _r).astype(dtype)
            assert float(np.abs(src).max()) * 2.0 ** _r < 32768, "split-f16 input bound violated"
            y = _act(PW(src.reshape(-1, cin), wt) + arr(sh_off, cout), act).reshape(n, h * w, cout).mean(axis=1).reshape(n, 1, 1, cout)
        elif kind == 21:     # pre-split pointwise + the next block's depthwise 3x3 / 1 / SAME (+ scale + shift + ReLU6), output as split rows
            a_log2, out_log2 = _r & 255, _r >> 8
            assert split_fmt.get(in_buf, 0) == a_log2, "op %d reads buffer %d in the wrong storage format" % (i, in_buf)
            assert cin % 32 == 0 and cout % 128 == 0 and h * w <= 288 and (oh * stride, ow * stride) == (h, w) and 0 < out_log2 <= 12
            assert (stride, pad_t, pad_l) == (1, 1, 1) or ((stride, pad_t, pad_l) == (2, 0, 0) and (h, w) in ((12, 12), (14, 14)) and act == 2)
            from hse_facerec_tf_amd.lowering import unsplit_pointwise_weights
            img = np.frombuffer(data, np.uint16, cout * cin * 2, w_off).reshape(cout, cin // 32, 64)
            wt = unsplit_pointwise_weights(img, np.frombuffer(data, np.float32, cout, sc_off), a_log2).astype(dtype)
            assert float(np.abs(src).max()) * 2.0 ** a_log2 < 32768, "split-f16 input bound violated"
            mid = _act(PW(src.reshape(-1, cin), wt) + arr(sh_off, cout), act).reshape(n, h, w, cout)
            dwc = arr(w2_off, 11 * cout).reshape(11, cout)
            amp = 2.0 ** out_log2
            xp = np.pad(mid, ((0, 0), (pad_t, 1), (pad_l, 1), (0, 0)))
            y = tfo.depthwise_conv2d(xp, dwc[:9].reshape(3, 3, cout, 1), (stride, stride), "VALID")
            y = np.minimum(np.maximum(y * (dwc[9] / amp) + dwc[10] / amp, 0), 6)
        elif kind in (15, 17):     # fused stem + the stride-2 depthwise of block 2 (17: with a declared input bound)
            from hse_facerec_tf_amd.lowering import unsplit_pointwise_weights
            pk = arr(w_off, 1952)
            k0 = pk[:864].reshape(3, 3, 3, 32)
            a_log2 = _r & 255
            if kind == 17:      # the conv kernel the device uses is the SPLIT image behind the fp32 pack: execute that one
                in_log2 = (_r >> 8) - 64
                cimg = np.frombuffer(data, np.uint16, 32 * 64, w_off + 1952 * 4).reshape(32, 1, 64)
                cds = np.frombuffer(data, np.float32, 32, w_off + (1952 + 1024) * 4)
                cw_t = unsplit_pointwise_weights(cimg, cds, in_log2)
                assert np.abs(cw_t[:, 27:]).max() == 0 and np.abs(cw_t[:, :27].T.reshape(3, 3, 3, 32) - k0).max() <= 2.0 ** -21 * np.abs(k0).max()
                k0 = cw_t[:, :27].T.reshape(3, 3, 3, 32).astype(dtype)
                assert float(np.abs(src).max()) < 2.0 ** (15 - in_log2), "declared input bound violated"
            h1, w1 = (h + 1) // 2, (w + 1) // 2
            pb = max((h1 - 1) * 2 + 3 - h - pad_t, 0)
            pr = max((w1 - 1) * 2 + 3 - w - pad_l, 0)
            c1 = _act(tfo.conv2d(src, k0, (2, 2), "", explicit_pads=(pad_t, pb, pad_l, pr)) + pk[864:896], 2)
            d1 = _act(tfo.depthwise_conv2d(np.pad(c1, ((0, 0), (1, 1), (1, 1), (0, 0))), pk[896:1184].reshape(3, 3, 32, 1), (1, 1), "VALID")
                      * pk[1184:1216] + pk[1216:1248], 2)
            img = np.frombuffer(data, np.uint16, 64 * 32 * 2, w2_off).reshape(64, 1, 64)
            ds = np.frombuffer(data, np.float32, 128, sh2_off)
            wt = unsplit_pointwise_weights(img, ds[:64], a_log2).astype(dtype)
            p1 = _act(PW(d1.reshape(-1, 32), wt) + ds[64:].astype(dtype), 2).reshape(n, h1, w1, 64)
            pt2, pl2 = (kw >> 4) & 1, (kw >> 5) & 1
            pb2 = max((oh - 1) * 2 + 3 - h1 - pt2, 0)
            pr2 = max((ow - 1) * 2 + 3 - w1 - pl2, 0)
            y = _act(tfo.depthwise_conv2d(np.pad(p1, ((0, 0), (pt2, pb2), (pl2, pr2), (0, 0))), pk[1248:1824].reshape(3, 3, 64, 1), (2, 2), "VALID")
                     * pk[1824:1888] + pk[1888:1952], act)
        elif kind == 14:     # fused stem: conv 3x3/2 (3->32) + shift + relu6 -> depthwise 3x3/1 -> pointwise 32->64 (split f16)
            from hse_facerec_tf_amd.lowering import unsplit_pointwise_weights
            pk = arr(w_off, 1248)
            k0 = pk[:864].reshape(3, 3, 3, 32)
            pb = max((oh - 1) * 2 + 3 - h - pad_t, 0)
            pr = max((ow - 1) * 2 + 3 - w - pad_l, 0)
            c1 = _act(tfo.conv2d(src, k0, (2, 2), "", explicit_pads=(pad_t, pb, pad_l, pr)) + pk[864:896], 2)
            mid = _act(tfo.depthwise_conv2d(np.pad(c1, ((0, 0), (1, 1), (1, 1), (0, 0))), pk[896:1184].reshape(3, 3, 32, 1), (1, 1), "VALID")
                       * pk[1184:1216] + pk[1216:1248], 2)
            img = np.frombuffer(data, np.uint16, cout * 32 * 2, w2_off).reshape(cout, 1, 64)
            ds = np.frombuffer(data, np.float32, 2 * cout, sh2_off)
            wt = unsplit_pointwise_weights(img, ds[:cout], _r).astype(dtype)
            y = _act(PW(mid.reshape(-1, 32), wt) + ds[cout:].astype(dtype), act).reshape(n, oh, ow, cout)
        elif kind == 4:
            y = src.mean(axis=(1, 2)).reshape(n, 1, 1, cin)
        elif kind == 5:
            k = arr(w_off, cin * cout).reshape(cin, cout)
            y = _act(src.reshape(n, cin).dot(k) + arr(sh_off, cout), act).reshape(n, 1, 1, cout)
        elif kind == 6:
            y = tfo.softmax(src.reshape(n, -1)).reshape(n, 1, 1, -1)
        elif kind == 11:     # fused depthwise (+scale+shift+relu6) -> pointwise (+shift+act)
            k = arr(w_off, 9 * cin).reshape(3, 3, cin, 1)
            pb = max((oh - 1) * stride + 3 - h - pad_t, 0)
            pr = max((ow - 1) * stride + 3 - w - pad_l, 0)
            xp = np.pad(src, ((0, 0), (pad_t, pb), (pad_l, pr), (0, 0)))
            mid = _act(tfo.depthwise_conv2d(xp, k, (stride, stride), "VALID") * arr(sc_off, cin) + arr(sh_off, cin), 2)
            wt = arr(w2_off, cin * cout).reshape(cout, cin)
            y = _act(PW(mid.reshape(-1, cin), wt) + arr(sh2_off, cout), act).reshape(n, oh, ow, cout)
        elif kind == 13:     # the same block with split-f16 pointwise weights (any channel count)
            from hse_facerec_tf_amd.lowering import unsplit_pointwise_weights
            k = arr(w_off, 9 * cin).reshape(3, 3, cin, 1)
            pb = max((oh - 1) * stride + 3 - h - pad_t, 0)
            pr = max((ow - 1) * stride + 3 - w - pad_l, 0)
            xp = np.pad(src, ((0, 0), (pad_t, pb), (pad_l, pr), (0, 0)))
            mid = _act(tfo.depthwise_conv2d(xp, k, (stride, stride), "VALID") * arr(sc_off, cin) + arr(sh_off, cin), 2)
            img = np.frombuffer(data, np.uint16, cout * cin * 2, w2_off).reshape(cout, cin // 32, 64)
            ds = np.frombuffer(data, np.float32, 2 * cout, sh2_off)
            wt = unsplit_pointwise_weights(img, ds[:cout], _r).astype(dtype)
            y = _act(PW(mid.reshape(-1, cin), wt) + ds[cout:].astype(dtype), act).reshape(n, oh, ow, cout)
        elif kind == 7:      # bf16 implicit-GEMM conv: weights [cout][kh*kw*cin] bf16, fp32 scale/shift, optional residual
            k = arr_bf16(w_off, kh * kw * cin * cout).reshape(cout, kh, kw, cin).transpose(1, 2, 3, 0)
            pb = max((oh - 1) * stride + kh - h - pad_t, 0)
            pr = max((ow - 1) * stride + kw - w - pad_l, 0)
            y = tfo.conv2d(src, k, (stride, stride), "", explicit_pads=(pad_t, pb, pad_l, pr))
            y = bf16_round(y * arr(sc_off, cout) + arr(sh_off, cout))
            if res is not None and w2_off != NO_OFFSET:
                # projected shortcut (lowering.fuse_proj): res_buf is the BLOCK INPUT, w2 its 1x1 kernel [cout][c2], shift2 = [scale2 | shift2],
                # the aux word c2 | stride2 << 12 | h2 << 14 | w2 << 23; the projection is rounded to bf16 where its tensor used to be stored
                c2, s2, h2, wd2 = _r & 0xFFF, (_r >> 12) & 3, (_r >> 14) & 0x1FF, (_r >> 23) & 0x1FF
                assert kh == 1 and kw == 1 and stride == 1
                k2 = arr_bf16(w2_off, c2 * cout).reshape(cout, 1, 1, c2).transpose(1, 2, 3, 0)
                ss2 = arr(sh2_off, 2 * cout)
                pr_ = tfo.conv2d(res.reshape(n, h2, wd2, c2), k2, (s2, s2), "", explicit_pads=(0, 0, 0, 0))
                y = y + bf16_round(pr_ * ss2[:cout] + ss2[cout:]).reshape(y.shape)
            elif res is not None and _r != 0:
                # strided residual (lowering.subsample_stage_tails): res_buf is a LARGER map [h2, w2, cout] read at every s2-th pixel
                s2, h2, wd2 = (_r >> 12) & 3, (_r >> 14) & 0x1FF, (_r >> 23) & 0x1FF
                assert (_r & 0xFFF) == 0 and kh == 1 and kw == 1 and stride == 1
                y = y + res.reshape(n, h2, wd2, cout)[:, ::s2, ::s2, :][:, :oh, :ow, :]
            elif res is not None:
                y = y + res.reshape(y.shape)
            y = bf16_round(_act(y, act))
        elif kind == 10:     # 7x7/2 pad-3 stem on the fp32 image: weights [64][8][32] bf16 (k = dy*32 + dx*3 + ci)
            wimg = arr_bf16(w_off, 64 * 256).reshape(64, 8, 32)
            assert np.all(wimg[:, 7] == 0) and np.all(wimg[:, :, 21:] == 0)
            k = wimg[:, :7, :21].reshape(64, 7, 7, 3).transpose(1, 2, 3, 0)
            pb = max((oh - 1) * 2 + 7 - h - 3, 0)
            pr = max((ow - 1) * 2 + 7 - w - 3, 0)
            y = tfo.conv2d(bf16_round(src), k, (2, 2), "", explicit_pads=(3, pb, 3, pr))
            y = bf16_round(_act(y * arr(sc_off, 64) + arr(sh_off, 64), act))
        elif kind == 20:     # kind 10 (ReLU) + kind 8 in one op: oh, ow are the POOLED size, _r = pool_pad_t | pool_pad_l << 4
            assert act == 1 and cin == 3 and cout == 64 and (_r & ~0x11) == 0
            wimg = arr_bf16(w_off, 64 * 256).reshape(64, 8, 32)
            k = wimg[:, :7, :21].reshape(64, 7, 7, 3).transpose(1, 2, 3, 0)
            ch, cw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            pb = max((ch - 1) * 2 + 7 - h - 3, 0)
            pr = max((cw - 1) * 2 + 7 - w - 3, 0)
            c1 = tfo.conv2d(bf16_round(src), k, (2, 2), "", explicit_pads=(3, pb, 3, pr))
            c1 = bf16_round(_act(c1 * arr(sc_off, 64) + arr(sh_off, 64), act))
            ppt, ppl = _r & 15, _r >> 4
            pb = max((oh - 1) * 2 + 3 - ch - ppt, 0)
            pr = max((ow - 1) * 2 + 3 - cw - ppl, 0)
            xp = np.pad(c1, ((0, 0), (ppt, pb), (ppl, pr), (0, 0)), constant_values=-np.inf)
            y = np.full((n, oh, ow, 64), -np.inf)
            for dy in range(3):
                for dx in range(3):
                    y = np.maximum(y, xp[:, dy:dy + 2 * (oh - 1) + 1:2, dx:dx + 2 * (ow - 1) + 1:2, :])
        elif kind == 8:      # 3x3/2 max-pool with clipped windows
            pb = max((oh - 1) * 2 + 3 - h - pad_t, 0)
            pr = max((ow - 1) * 2 + 3 - w - pad_l, 0)
            xp = np.pad(src, ((0, 0), (pad_t, pb), (pad_l, pr), (0, 0)), constant_values=-np.inf)
            y = np.full((n, oh, ow, cin), -np.inf)
            for dy in range(3):
                for dx in range(3):
                    y = np.maximum(y, xp[:, dy:dy + 2 * (oh - 1) + 1:2, dx:dx + 2 * (ow - 1) + 1:2, :])
        elif kind == 9:
            y = src.mean(axis=(1, 2)).reshape(n, 1, 1, cin)
        elif kind == 18:     # general fp32 conv: HWIO kernel, optional per-channel scale / shift, optional residual, act
            k = arr(w_off, kh * kw * cin * cout).reshape(kh, kw, cin, cout)
            pb = max((oh - 1) * stride + kh - h - pad_t, 0)
            pr = max((ow - 1) * stride + kw - w - pad_l, 0)
            y = tfo.conv2d(src, k, (stride, stride), "", explicit_pads=(pad_t, pb, pad_l, pr))
            if sc_off != NO_OFFSET:
                y = y * arr(sc_off, cout)
            if sh_off != NO_OFFSET:
                y = y + arr(sh_off, cout)
            if res is not None and _r != 0:      # strided residual, as kind 7's
                s2, h2, wd2 = (_r >> 12) & 3, (_r >> 14) & 0x1FF, (_r >> 23) & 0x1FF
                assert (_r & 0xFFF) == 0 and kh == 1 and kw == 1 and stride == 1 and cout % 64 == 0
                y = y + res.reshape(n, h2, wd2, cout)[:, ::s2, ::s2, :][:, :oh, :ow, :]
            elif res is not None:
                y = y + res.reshape(y.shape)
            y = _act(y, act)
        elif kind == 19:     # k x k / stride max-pool with clipped windows, fp32
            assert kh == kw
            pb = max((oh - 1) * stride + kh - h - pad_t, 0)
            pr = max((ow - 1) * stride + kh - w - pad_l, 0)
            xp = np.pad(src, ((0, 0), (pad_t, pb), (pad_l, pr), (0, 0)), constant_values=-np.inf)
            y = np.full((n, oh, ow, cin), -np.inf)
            for dy in range(kh):
                for dx in range(kh):
                    y = np.maximum(y, xp[:, dy:dy + stride * (oh - 1) + 1:stride, dx:dx + stride * (ow - 1) + 1:stride, :])
        else:
            raise AssertionError("unknown op kind %d" % kind)
        assert y.shape[1:] == (oh, ow, cout), (i, y.shape, (oh, ow, cout))
        if kind not in (12, 16, 21, 22):
            assert split_fmt.get(in_buf, 0) == 0 or in_buf == -1, "op %d (kind %d) reads split rows it cannot decode" % (i, kind)
        if sub2:
            in_registers, y = y, y.reshape(n, h, w, cout)[:, ::2, ::2, :]
        mem[out_buf] = y
        split_fmt[out_buf] = _r if kind == 2 else (_r >> 8 if kind == 21 else 0)
    outs = {}
    for slot, name in enumerate(("features", "age_probs", "gender")):
        if p["out_buf"][slot] >= 0:
            outs[name] = mem[p["out_buf"][slot]].reshape(n, -1)[:, :p["out_elems"][slot]]
    return outs
