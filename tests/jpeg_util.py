"""A minimal baseline JPEG ENCODER and an independent numpy restatement of the decode pipeline, for testing the native driver's
JPEG reader (easysfm_amd/host/esfm_jpeg.hpp) without any imaging library in the image.

encode(): JFIF baseline, 8-bit, YCbCr 4:4:4 / 4:2:2 / 4:2:0 or gray, the Annex-K Huffman tables, optional restart interval.
It also returns the quantised coefficient blocks it wrote, so reference_decode() can rebuild the expected pixels from THEM with
libjpeg's published arithmetic (islow IDCT, fancy upsampling, 16-bit colour tables) written a second time, vectorised -- a check
of the C++ reader's entropy decoding and of its integer pipeline, bit for bit.
"""
import numpy as np

ZZ = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
QL = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
               18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
QC = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)
DC_L = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_C = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_L = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
        [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
         0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
         0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
         0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
         0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
         0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
         0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
AC_C = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
        [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
         0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
         0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
         0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
         0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
         0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
         0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])


def _codes(bits, vals):
    out, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            out[vals[k]] = (code, ln)
            code += 1; k += 1
        code <<= 1
    return out


class _Bits:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, code, ln):
        self.acc = (self.acc << ln) | (code & ((1 << ln) - 1)); self.n += ln
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _dct_matrix():
    m = np.zeros((8, 8))
    for k in range(8):
        for x in range(8):
            m[k, x] = (np.sqrt(0.125) if k == 0 else 0.5) * np.cos((2 * x + 1) * k * np.pi / 16)
    return m


def _scaled(q, quality):
    s = 5000 / quality if quality < 50 else 200 - 2 * quality
    return np.clip((q * s + 50) // 100, 1, 255).astype(np.int64)


def encode(rgb, subsampling="420", quality=90, restart=0):
    """rgb: HxWx3 uint8 (or HxW gray).  Returns (jpeg bytes, info) with info = dict(coefs=[per component (bh, bw, 64) ints in
    NATURAL order, already multiplied by the quantiser], h=[...], v=[...], width, height)."""
    gray = rgb.ndim == 2
    H, W = rgb.shape[:2]
    if gray:
        planes = [rgb.astype(np.float64)]
        hv = [(1, 1)]
    else:
        r, g, b = [rgb[:, :, i].astype(np.float64) for i in range(3)]
        y = 0.299 * r + 0.587 * g + 0.114 * b
        cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128
        cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128
        planes = [y, cb, cr]
        hv = {"444": [(1, 1)] * 3, "422": [(2, 1), (1, 1), (1, 1)], "420": [(2, 2), (1, 1), (1, 1)]}[subsampling]
    hmax, vmax = max(h for h, _ in hv), max(v for _, v in hv)
    mcux, mcuy = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    D = _dct_matrix()
    qts = [_scaled(QL, quality)] + ([_scaled(QC, quality)] * 2 if not gray else [])
    blocks = []
    for ci, (pl, (h, v)) in enumerate(zip(planes, hv)):
        fx, fy = hmax // h, vmax // v
        dw, dh = -(-W * h // hmax), -(-H * v // vmax)
        ph, pw = -(-H // fy) * fy, -(-W // fx) * fx
        p = np.pad(pl, ((0, ph - H), (0, pw - W)), mode="edge")
        p = p.reshape(ph // fy, fy, pw // fx, fx).mean(axis=(1, 3))[:dh, :dw]
        bw, bh = mcux * h, mcuy * v
        p = np.pad(p, ((0, bh * 8 - dh), (0, bw * 8 - dw)), mode="edge") - 128.0
        blk = p.reshape(bh, 8, bw, 8).transpose(0, 2, 1, 3)
        co = np.einsum("ki,abij,lj->abkl", D, blk, D).reshape(bh, bw, 64)
        q = qts[ci].reshape(1, 1, 64)
        blocks.append(np.rint(co / q).astype(np.int64))
    # entropy coding
    tabs = [(_codes(*DC_L), _codes(*AC_L))] + ([(_codes(*DC_C), _codes(*AC_C))] * 2 if not gray else [])
    bits = _Bits()
    pred = [0] * len(planes)
    scan = bytearray()
    count = 0
    rst = 0

    def category(v):
        return int(abs(v)).bit_length()

    for my in range(mcuy):
        for mx in range(mcux):
            if restart and count == restart:
                bits.flush(); scan += bits.out; scan += bytes([0xFF, 0xD0 + rst]); rst = (rst + 1) & 7
                bits.__init__(); pred = [0] * len(planes); count = 0
            for ci, (h, v) in enumerate(hv):
                dct, act = tabs[ci]
                for by in range(v):
                    for bx in range(h):
                        zq = blocks[ci][my * v + by, mx * h + bx][ZZ]
                        diff = int(zq[0]) - pred[ci]; pred[ci] = int(zq[0])
                        t = category(diff)
                        bits.put(*dct[t])
                        if t:
                            bits.put(diff if diff >= 0 else diff + (1 << t) - 1, t)
                        run = 0
                        last = np.flatnonzero(zq[1:])
                        last = int(last[-1]) + 1 if last.size else 0
                        for k in range(1, last + 1):
                            c = int(zq[k])
                            if c == 0:
                                run += 1; continue
                            while run > 15:
                                bits.put(*act[0xF0]); run -= 16
                            s = category(c)
                            bits.put(*act[(run << 4) | s]); bits.put(c if c >= 0 else c + (1 << s) - 1, s); run = 0
                        if last < 63:
                            bits.put(*act[0x00])
            count += 1
    bits.flush(); scan += bits.out

    def seg(marker, payload):
        return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)

    out = bytearray(b"\xFF\xD8")
    out += seg(0xE0, b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    out += seg(0xDB, bytes([0]) + bytes(int(x) for x in qts[0][ZZ]))
    if not gray:
        out += seg(0xDB, bytes([1]) + bytes(int(x) for x in qts[1][ZZ]))
    sof = bytes([8]) + H.to_bytes(2, "big") + W.to_bytes(2, "big") + bytes([len(planes)])
    for ci, (h, v) in enumerate(hv):
        sof += bytes([ci + 1, (h << 4) | v, 0 if ci == 0 else 1])
    out += seg(0xC0, sof)
    for tc_th, (b_, v_) in ([(0x00, DC_L), (0x10, AC_L)] + ([(0x01, DC_C), (0x11, AC_C)] if not gray else [])):
        out += seg(0xC4, bytes([tc_th]) + bytes(b_) + bytes(v_))
    if restart:
        out += seg(0xDD, restart.to_bytes(2, "big"))
    sos = bytes([len(planes)])
    for ci in range(len(planes)):
        sos += bytes([ci + 1, 0x00 if ci == 0 else 0x11])
    out += seg(0xDA, sos + bytes([0, 63, 0])) + scan + b"\xFF\xD9"
    info = dict(coefs=[blocks[ci] * qts[ci].reshape(1, 1, 64) for ci in range(len(planes))], h=[h for h, _ in hv], v=[v for _, v in hv],
                width=W, height=H)
    return bytes(out), info


def _idct_islow(co):
    """libjpeg jidctint.c on an array of blocks (..., 64) of dequantised coefficients in natural order -> (..., 8, 8) uint8."""
    CB, P1 = 13, 2
    F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, l=25172)

    def desc(x, n):
        return (x + (1 << (n - 1))) >> n

    def pass1d(v, shift_in, n):
        # v: (..., 8) along the transformed axis
        z2, z3 = v[..., 2], v[..., 6]
        z1 = (z2 + z3) * F["c"]
        t2 = z1 - z3 * F["h"]; t3 = z1 + z2 * F["d"]
        t0 = (v[..., 0] + v[..., 4]) << CB; t1 = (v[..., 0] - v[..., 4]) << CB
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        t0, t1, t2, t3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
        z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
        z5 = (z3 + z4) * F["f"]
        t0, t1, t2, t3 = t0 * F["a"], t1 * F["j"], t2 * F["l"], t3 * F["g"]
        z1, z2, z3, z4 = -z1 * F["e"], -z2 * F["k"], -z3 * F["i"] + z5, -z4 * F["b"] + z5
        t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
        o = np.stack([t10 + t3, t11 + t2, t12 + t1, t13 + t0, t13 - t0, t12 - t1, t11 - t2, t10 - t3], axis=-1)
        return desc(o, n)

    b = co.reshape(co.shape[:-1] + (8, 8)).astype(np.int64)          # [row][col]
    ws = np.swapaxes(pass1d(np.swapaxes(b, -1, -2), 0, CB - P1), -1, -2)   # columns first
    out = pass1d(ws, 0, CB + P1 + 3) + 128
    return np.clip(out, 0, 255).astype(np.uint8)


def reference_decode(info):
    """Expected BGR image (HxWx3 uint8) of a file written by encode(), from its coefficient blocks."""
    W, H = info["width"], info["height"]
    hmax, vmax = max(info["h"]), max(info["v"])
    full = []
    for co, h, v in zip(info["coefs"], info["h"], info["v"]):
        bh, bw, _ = co.shape
        px = _idct_islow(co).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8).astype(np.int64)
        dw, dh = -(-W * h // hmax), -(-H * v // vmax)
        px = px[:dh, :dw]
        hs, vs = hmax // h, vmax // v
        if hs == 1 and vs == 1:
            up = px
        elif hs == 2 and vs == 1:
            left = np.concatenate([px[:, :1], px[:, :-1]], axis=1); right = np.concatenate([px[:, 1:], px[:, -1:]], axis=1)
            even = (3 * px + left + 1) >> 2; odd = (3 * px + right + 2) >> 2
            even[:, 0] = px[:, 0]; odd[:, -1] = px[:, -1]
            up = np.empty((dh, 2 * dw), np.int64); up[:, 0::2] = even; up[:, 1::2] = odd
        elif hs == 2 and vs == 2:
            above = np.concatenate([px[:1], px[:-1]], axis=0); below = np.concatenate([px[1:], px[-1:]], axis=0)
            rows = np.empty((2 * dh, dw), np.int64); rows[0::2] = 3 * px + above; rows[1::2] = 3 * px + below      # column sums
            left = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1); right = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
            even = (3 * rows + left + 8) >> 4; odd = (3 * rows + right + 7) >> 4
            even[:, 0] = (rows[:, 0] * 4 + 8) >> 4; odd[:, -1] = (rows[:, -1] * 4 + 7) >> 4
            up = np.empty((2 * dh, 2 * dw), np.int64); up[:, 0::2] = even; up[:, 1::2] = odd
        else:
            up = np.repeat(np.repeat(px, vs, axis=0), hs, axis=1)
        full.append(up[:H, :W])
    if len(full) == 1:
        return np.repeat(full[0][:, :, None], 3, axis=2).astype(np.uint8)
    y, cb, cr = full
    x = np.arange(256, dtype=np.int64) - 128
    cr_r = (91881 * x + 32768) >> 16; cb_b = (116130 * x + 32768) >> 16
    cr_g = -46802 * x; cb_g = -22554 * x + 32768
    r = np.clip(y + cr_r[cr], 0, 255); g = np.clip(y + ((cb_g[cb] + cr_g[cr]) >> 16), 0, 255); b = np.clip(y + cb_b[cb], 0, 255)
    return np.stack([b, g, r], axis=2).astype(np.uint8)
