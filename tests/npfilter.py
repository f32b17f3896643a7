"""A second, independent statement of the depth-heuristic prefilter (project_cloud.cu:28-187,
331-392) in vectorised numpy, written from the reference lines without looking at
oracle/rtr_oracle.c's structure: whole-array float32 operations (numpy rounds every float32
operation once; the fused multiply-adds of the contract are formed in float64, whose 53-bit
product of two float32 values is exact, then rounded once to float32).  Agreement with the C
oracle pins the oracle's filter the way tests/pymodel.py pins its projection."""
import numpy as np

F = np.float32
EMPTY = np.uint32(0x7F7FFFFF)


def _fma(a, b, c):  # round32(a*b + c): a*b exact in float64
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(F) if np.isscalar(b) else \
        (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def reduce2(hi):  # project_cloud.cu:28-53
    h, w = hi.shape[0] // 2, hi.shape[1] // 2
    p0, p1 = hi[0:2 * h:2, 0:2 * w:2], hi[0:2 * h:2, 1:2 * w:2]
    p2, p3 = hi[1:2 * h:2, 0:2 * w:2], hi[1:2 * h:2, 1:2 * w:2]
    with np.errstate(invalid="ignore"):
        l0 = np.where(p0 < p1, p0, p1)
        l1 = np.where(p2 < p3, p2, p3)
        return np.where(l0 < l1, l0, l1).astype(F)


def laplacian(img, thr):  # project_cloud.cu:55-79, on the (possibly truncated) view passed in
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    if h < 3 or w < 3:
        return out
    k = [0, 1, 0, 1, -4, 1, 0, 1, 0]
    s = np.zeros((h - 2, w - 2), F)
    with np.errstate(all="ignore"):
        c = 0
        for ky in range(3):
            for kx in range(3):
                s = _fma(img[ky:ky + h - 2, kx:kx + w - 2], F(k[c]), s)
                c += 1
        out[1:-1, 1:-1] = np.where(s > F(thr), 255, 0)
    return out


def compare(lo, hi, grad, strength):  # project_cloud.cu:88-126; lo: (lh, lw), hi: (2lh, 2lw)
    lh, lw = lo.shape
    pad = np.full((lh + 2, lw + 2), F(-1.0), F)  # getPixelValue: out of bounds -> -1
    pad[1:-1, 1:-1] = lo
    with np.errstate(all="ignore"):
        prod = (pad * F(strength)).astype(F)
    up = lambda a: np.repeat(np.repeat(a, 2, axis=0), 2, axis=1)  # noqa: E731  parent of (x, y) is (x/2, y/2)
    with np.errstate(invalid="ignore"):
        centre = hi <= up(prod[1:-1, 1:-1])
        anyn = np.zeros(hi.shape, bool)
        for dy in range(3):
            for dx in range(3):
                anyn |= hi <= up(prod[dy:dy + lh, dx:dx + lw])
        keep = np.where(up(grad) > 0, anyn, centre)
        keep &= ~(hi.astype(np.float64) >= 3.4028e38)
    return np.where(keep, 255, 0).astype(np.uint8)


def resize_into(lo, hi, mask):  # project_cloud.cu:128-161, in place where mask == 0
    oh, ow = hi.shape
    lh, lw = lo.shape
    x = np.arange(ow, dtype=F)
    y = np.arange(oh, dtype=F)
    inx = ((x + F(0.5)) / F(2.0) - F(0.5)).astype(F)
    iny = ((y + F(0.5)) / F(2.0) - F(0.5)).astype(F)
    x0 = np.floor(inx).astype(np.int64)
    y0 = np.floor(iny).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = np.clip(x0, 0, lw - 1), np.clip(x1, 0, lw - 1)
    y0, y1 = np.clip(y0, 0, lh - 1), np.clip(y1, 0, lh - 1)
    wx = (inx - x0.astype(F)).astype(F)[None, :]
    wy = (iny - y0.astype(F)).astype(F)[:, None]
    with np.errstate(all="ignore"):
        def row(yy):
            a, b = lo[yy][:, x0], lo[yy][:, x1]
            return _fma(np.broadcast_to(wx, a.shape), b, ((F(1) - wx).astype(F) * a).astype(F))
        v0, v1 = row(y0), row(y1)
        out = _fma(np.broadcast_to(wy, v0.shape), v1, ((F(1) - wy).astype(F) * v0).astype(F))
    hi[mask == 0] = out[mask == 0]


def apply_filter(depth_bits, img, strength=1.025, thr=0.03, levels=4):
    """-> dict(depth f32, img, mask, tensor u16 [5,H,W], minmax u32[2]) under the rules of
    DESIGN.md (rows >= H_eff: mask = non-empty; tensor plane stride W*H)."""
    H, W = depth_bits.shape
    lv = [depth_bits.view(F).copy()]
    for i in range(1, levels + 1):
        lv.append(reduce2(lv[i - 1]))
    ch, cw = lv[levels].shape
    mask = None
    for i in range(levels, 0, -1):
        grad = laplacian(lv[i][:ch, :cw], thr)
        ch, cw = ch * 2, cw * 2
        mask = compare(lv[i][:ch // 2, :cw // 2], lv[i - 1][:ch, :cw], grad, strength)
        if i > 1:
            sub = lv[i - 1][:ch, :cw]
            resize_into(lv[i][:ch // 2, :cw // 2], sub, mask)
    bits = depth_bits[:ch, :cw]
    valid = bits[bits != EMPTY]
    mn = np.uint32(valid.min()) if valid.size else np.uint32(0xFFFFFFFF)
    mx = np.uint32(valid.max()) if valid.size else np.uint32(0)
    full = np.zeros((H, W), np.uint8)
    full[:ch, :cw] = mask
    d = depth_bits.view(F).copy()
    full[ch:] = np.where(d[ch:].astype(np.float64) >= 3.4028e38, 0, 255)
    keep = full > 0
    out_img = np.where(keep[..., None], img, 0).astype(np.uint8)
    tensor = np.zeros((5, H, W), np.uint16)
    with np.errstate(all="ignore"):
        fmn = np.array([mn], np.uint32).view(F)[0]
        rng = F(np.array([mx], np.uint32).view(F)[0] - fmn)
        for c in range(3):
            v = (img[..., c].astype(F).astype(np.float16).astype(F) / F(255.0)).astype(F).astype(np.float16)
            tensor[c] = np.where(keep, v.view(np.uint16), 0)
        tensor[3] = np.where(keep, 0x3C00, 0)
        dd = (((d - fmn).astype(F)).astype(np.float16).astype(F) / rng).astype(F).astype(np.float16).view(np.uint16)
        dd = np.where(np.isnan(dd.view(np.float16)), np.uint16(0x7E00), dd)
        tensor[4] = np.where(keep, dd, 0xBC00)
    d[~keep] = F(-1.0)
    return {"depth": d, "img": out_img, "mask": full, "tensor": tensor, "minmax": np.array([mn, mx], np.uint32)}
