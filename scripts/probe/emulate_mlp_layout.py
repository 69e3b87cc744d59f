"""Index-logic emulation of csrc/nerf_mlp_bf16.hip in float64 (no rounding): the MFMA operand/result layouts and the
ds_read_b64_tr_b16 mapping are the ones scripts/probe/probe_mfma.hip confirmed on gfx950.  Checks that the chained K
ordering, the permuted weight images, the transposition images and the dW/db result mappings reproduce plain matrix
algebra for one 32-sample tile."""
import numpy as np

IN, H, X2, PE = 32, 64, 42, 27
ONES = 16 + PE
rng = np.random.default_rng(0)


def phi16(p): return 8 * ((p & 7) >> 2) + 4 * (p >> 3) + (p & 3)
def phi(s): return (s & ~15) + phi16(s & 15)
def acc_row(r, g): return (r & 3) + 8 * (r >> 2) + 4 * g


def mma32(A_lane, B_lane, C_lane):
    """A_lane[l][8], B_lane[l][8], C_lane[l][16] -> D_lane[l][16]"""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        i, g = l & 31, l >> 5
        A[i, 8 * g:8 * g + 8] = A_lane[l]; B[8 * g:8 * g + 8, i] = B_lane[l]
    D = A @ B
    out = np.array(C_lane, dtype=float).copy()
    for l in range(64):
        n, g = l & 31, l >> 5
        for r in range(16): out[l, r] += D[acc_row(r, g), n]
    return out


def mma16(A_lane, B_lane, C_lane):
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        i, kg = l & 15, l >> 4
        A[i, 8 * kg:8 * kg + 8] = A_lane[l]; B[8 * kg:8 * kg + 8, i] = B_lane[l]
    D = A @ B
    out = np.array(C_lane, dtype=float).copy()
    for l in range(64):
        c, rg = l & 15, l >> 4
        for rr in range(4): out[l, rr] += D[4 * rg + rr, c]
    return out


# ---- parameters
W1 = rng.normal(size=(H, IN)); b1 = rng.normal(size=H); W2 = rng.normal(size=(16, H)); b2 = rng.normal(size=16)
W3 = rng.normal(size=(H, X2)); b3 = rng.normal(size=H); W4 = rng.normal(size=(H, H)); b4 = rng.normal(size=H)
W5 = rng.normal(size=(3, H)); b5 = rng.normal(size=3)
W1 *= .3; W2 *= .2; W3 *= .2; W4 *= .2; W5 *= .2

# ---- LDS images (as 2-D arrays [row][slot])
W1p = W1.copy()
W2p = np.array([[W2[r, phi(s)] for s in range(64)] for r in range(16)])
W3p = np.zeros((64, 48))
for r in range(64):
    for s in range(48):
        if s < 16:
            m = phi16(s); W3p[r, s] = W3[r, m - 1] if m else 0.0
        elif s < ONES: W3p[r, s] = W3[r, s - 1]
        elif s == ONES: W3p[r, s] = b3[r]
W4p = np.array([[W4[r, phi(s)] for s in range(64)] for r in range(64)])
W5p = np.zeros((4, 64)); W5p[:3] = [[W5[r, phi(s)] for s in range(64)] for r in range(3)]
W5T = np.array([[W5[p, k] if p < 3 else 0.0 for p in range(16)] for k in range(64)])
W4T = np.array([[W4[phi(s), k] for s in range(64)] for k in range(64)])
W3T = np.array([[W3[phi(s), m - 1] if m else 0.0 for s in range(64)] for m in range(16)])
W2T = np.array([[W2[phi16(p), k] for p in range(16)] for k in range(64)])
W1T = np.array([[W1[phi(s), k] for s in range(64)] for k in range(32)])

lanes = np.arange(64); N = lanes & 31; G = lanes >> 5


def a_op(img, rows, kb): return np.array([img[rows[l], 16 * kb + 8 * G[l]:16 * kb + 8 * G[l] + 8] for l in range(64)])
def pack(acc, base, relu): v = acc[:, base:base + 8]; return np.maximum(v, 0) if relu else v.copy()


# ---- inputs of one tile
x0 = rng.normal(size=(32, IN)); dirs = rng.normal(size=(32, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
g_rgb = rng.normal(size=(32, 3)); g_den = rng.normal(size=32)


def encode(d):
    out = [d[0], d[1], d[2]]
    for k in range(4): out += [np.sin(2 ** k * d[a]) for a in range(3)]
    for k in range(4): out += [np.cos(2 ** k * d[a]) for a in range(3)]
    return np.array(out)


# ---- plain reference (float64)
pe = np.array([encode(d) for d in dirs])
h1 = np.maximum(x0 @ W1.T + b1, 0); y = h1 @ W2.T + b2
x2r = np.concatenate([y[:, 1:], pe], 1); h2 = np.maximum(x2r @ W3.T + b3, 0); h3 = np.maximum(h2 @ W4.T + b4, 0)
sg = 1 / (1 + np.exp(-(h3 @ W5.T + b5)))
dy5 = g_rgb * sg * (1 - sg); dh3 = (dy5 @ W5) * (h3 > 0); dh2 = (dh3 @ W4) * (h2 > 0); dx2 = dh2 @ W3
dy2 = np.concatenate([(g_den * (y[:, 0] > 0))[:, None], dx2[:, :15]], 1); dh1 = (dy2 @ W2) * (h1 > 0); dx0 = dh1 @ W1
ref = dict(dW5=dy5.T @ h3, db5=dy5.sum(0), dW4=dh3.T @ h2, db4=dh3.sum(0), dW3=dh2.T @ x2r, db3=dh2.sum(0),
           dW2=dy2.T @ h1, db2=dy2.sum(0), dW1=dh1.T @ x0, db1=dh1.sum(0))

# ---- emulated kernel
X0 = [np.array([x0[N[l], 16 * kb + 8 * G[l]:16 * kb + 8 * G[l] + 8] for l in range(64)]) for kb in range(2)]
bias = lambda b, t: np.array([[b[32 * t + acc_row(r, G[l])] for r in range(16)] for l in range(64)])
Hh1 = [None] * 4
for t in range(2):
    acc = bias(b1, t)
    for kb in range(2): acc = mma32(a_op(W1p, 32 * t + N, kb), X0[kb], acc)
    Hh1[2 * t] = pack(acc, 0, True); Hh1[2 * t + 1] = pack(acc, 8, True)
acc = np.zeros((64, 16))
for kb in range(4): acc = mma32(a_op(W2p, N & 15, kb), Hh1[kb], acc)
for l in range(64):
    for r in range(8): acc[l, r] += b2[acc_row(r, G[l])]
y0 = acc[:, 0].copy()
acc[G == 0, 0] = 0.0
X2p = [pack(acc, 0, False), None, None]
for kb in (1, 2):
    blk = np.zeros((64, 8))
    for l in range(64):
        for j in range(8):
            e = 16 * (kb - 1) + 8 * G[l] + j
            blk[l, j] = pe[N[l], e] if e < PE else (1.0 if e == PE else 0.0)
    X2p[kb] = blk
Hh2 = [None] * 4; Hh3 = [None] * 4
for t in range(2):
    acc = np.zeros((64, 16))
    for kb in range(3): acc = mma32(a_op(W3p, 32 * t + N, kb), X2p[kb], acc)
    Hh2[2 * t] = pack(acc, 0, True); Hh2[2 * t + 1] = pack(acc, 8, True)
for t in range(2):
    acc = bias(b4, t)
    for kb in range(4): acc = mma32(a_op(W4p, 32 * t + N, kb), Hh2[kb], acc)
    Hh3[2 * t] = pack(acc, 0, True); Hh3[2 * t + 1] = pack(acc, 8, True)
acc = np.zeros((64, 16))
for kb in range(4): acc = mma32(a_op(W5p, np.minimum(N, 3), kb), Hh3[kb], acc)
SG = 1 / (1 + np.exp(-(acc[:, :3] + b5)))
assert np.allclose(SG[:32], sg), "forward rgb"
assert np.allclose(y0[:32], y[:, 0]), "forward y0"

masked = lambda acc, base, h: acc[:, base:base + 8] * (h > 0)
DY5 = np.zeros((64, 8))
for l in range(32): DY5[l, :3] = g_rgb[l] * SG[l] * (1 - SG[l])
DH3 = [None] * 4
for t in range(2):
    acc = mma32(a_op(W5T, 32 * t + N, 0), DY5, np.zeros((64, 16)))
    DH3[2 * t] = masked(acc, 0, Hh3[2 * t]); DH3[2 * t + 1] = masked(acc, 8, Hh3[2 * t + 1])
DH2 = [None] * 4
for t in range(2):
    acc = np.zeros((64, 16))
    for kb in range(4): acc = mma32(a_op(W4T, 32 * t + N, kb), DH3[kb], acc)
    DH2[2 * t] = masked(acc, 0, Hh2[2 * t]); DH2[2 * t + 1] = masked(acc, 8, Hh2[2 * t + 1])
acc = np.zeros((64, 16))
for kb in range(4): acc = mma32(a_op(W3T, N & 15, kb), DH2[kb], acc)
for l in range(32): acc[l, 0] = g_den[l] if y0[l] > 0 else 0.0
DY2 = pack(acc, 0, False)
DH1 = [None] * 4
for t in range(2):
    acc = mma32(a_op(W2T, 32 * t + N, 0), DY2, np.zeros((64, 16)))
    DH1[2 * t] = masked(acc, 0, Hh1[2 * t]); DH1[2 * t + 1] = masked(acc, 8, Hh1[2 * t + 1])
acc = np.zeros((64, 16))
for kb in range(4): acc = mma32(a_op(W1T, N, kb), DH1[kb], acc)
DX0 = np.zeros((32, 32))
for l in range(64):
    for q in range(4):
        for i in range(4): DX0[N[l], 8 * q + 4 * G[l] + i] = acc[l, 4 * q + i]
print("dX0 max err", np.abs(DX0 - dx0).max())
assert np.allclose(DX0, dx0), "grad_feats"

# ---- transposition images (element = one value; byte offsets / 2) and the transposing reads
REGION = 320   # elements


def new_img(): return np.full(8 * REGION, np.nan)
def store_chained(img, kb, P):
    for l in range(64):
        base = (2 * N[l] + G[l]) * 4
        img[(2 * kb) * REGION + base:(2 * kb) * REGION + base + 4] = P[l, :4]
        img[(2 * kb + 1) * REGION + base:(2 * kb + 1) * REGION + base + 4] = P[l, 4:]
def store_natural(img, kb, P):
    for l in range(64):
        base = G[l] * REGION + N[l] * 8 + 2 * kb * REGION
        img[base:base + 8] = P[l]
def tr_read(img, elem_off):
    """ds_read_b64_tr_b16: lane l element j <- element (l%16)%4 of the chunk addressed by lane 16*(l//16) + 4*j + (l%16)//4"""
    out = np.zeros((64, 4))
    for l in range(64):
        for j in range(4):
            src = 16 * (l // 16) + 4 * j + (l % 16) // 4
            out[l, j] = img[elem_off[src] + (l % 16) % 4]
    return out
def load_transposed(img, fb):
    off = np.array([((l >> 1) & 1) * REGION + (8 * (l >> 4) + 2 * ((l >> 2) & 3) + (l & 1)) * 4 + fb * 2 * REGION for l in range(64)])
    return np.concatenate([tr_read(img, off), tr_read(img, off + 128)], 1)


def dw(dy_img, x_img, nit, nkt):
    blocks = {}
    for it in range(nit):
        a = load_transposed(dy_img, it)
        blocks[(it, 'b')] = mma16(a, np.ones((64, 8)), np.zeros((64, 4)))
        for kt in range(nkt): blocks[(it, kt)] = mma16(a, load_transposed(x_img, kt), np.zeros((64, 4)))
    return blocks
def assemble(blocks, nit, nkt):
    M = np.zeros((16 * nit, 16 * nkt)); b = np.zeros(16 * nit)
    for l in range(64):
        c, rg = l & 15, l >> 4
        for rr in range(4):
            for it in range(nit):
                for kt in range(nkt): M[16 * it + 4 * rg + rr, 16 * kt + c] = blocks[(it, kt)][l, rr]
                if c == 0: b[16 * it + 4 * rg + rr] = blocks[(it, 'b')][l, rr]
    return M, b


imgY, imgX = new_img(), new_img()
store_natural(imgY, 0, DY5); [store_chained(imgX, kb, Hh3[kb]) for kb in range(4)]
M, b = assemble(dw(imgY, imgX, 1, 4), 1, 4)
assert np.allclose(M[:3], ref['dW5']) and np.allclose(b[:3], ref['db5']), "dW5"
imgY, imgX = new_img(), new_img()
[store_chained(imgY, kb, DH3[kb]) for kb in range(4)]; [store_chained(imgX, kb, Hh2[kb]) for kb in range(4)]
M, b = assemble(dw(imgY, imgX, 4, 4), 4, 4)
assert np.allclose(M, ref['dW4']) and np.allclose(b, ref['db4']), "dW4"
imgY, imgX = new_img(), new_img()
[store_chained(imgY, kb, DH2[kb]) for kb in range(4)]
store_chained(imgX, 0, X2p[0]); store_natural(imgX, 1, X2p[1]); store_natural(imgX, 2, X2p[2])
M, b = assemble(dw(imgY, imgX, 4, 3), 4, 3)
dW3 = np.zeros((64, X2)); db3 = np.zeros(64)
for u in range(48):
    if u < 16:
        if u >= 1: dW3[:, u - 1] = M[:, u]
    elif u < ONES: dW3[:, u - 1] = M[:, u]
    elif u == ONES: db3 = M[:, u]
assert np.allclose(dW3, ref['dW3']) and np.allclose(db3, ref['db3']), "dW3"
imgY, imgX = new_img(), new_img()
store_chained(imgY, 0, DY2); [store_chained(imgX, kb, Hh1[kb]) for kb in range(4)]
M, b = assemble(dw(imgY, imgX, 1, 4), 1, 4)
assert np.allclose(M, ref['dW2']) and np.allclose(b, ref['db2']), "dW2"
imgY, imgX = new_img(), new_img()
[store_chained(imgY, kb, DH1[kb]) for kb in range(4)]; store_natural(imgX, 0, X0[0]); store_natural(imgX, 1, X0[1])
M, b = assemble(dw(imgY, imgX, 4, 2), 4, 2)
assert np.allclose(M, ref['dW1']) and np.allclose(b, ref['db1']), "dW1"
print("layout emulation: all checks passed")
