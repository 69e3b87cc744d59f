"""Can the decoder backward read its transposed weight operands out of the FORWARD weight images (dropping the 22 KB of
transposed copies in LDS)?  Index-logic check in float64, with the ds_read_b64_tr_b16 lane / element mapping that
scripts/probe/probe_mfma.hip confirmed on gfx950: result element j of lane i of a 16-lane group = element i % 4 of the 8-byte chunk
addressed by lane 4 j + i / 4 of that group.

A operand of a backward block (v_mfma_f32_32x32x16_bf16, lane l: row m = l & 31, K slots 8 g .. 8 g + 7 of block kb, g = l >> 5):
today one ds_read_b128 from the transposed image `WT[32 t + m][16 kb + 8 g + 0..7]`; here two transposing reads from the forward
image, lane l supplying the chunk address
    row = 16 kb + 4 g (+ 8 for the second read) + (l % 16) / 4,   column = (column block of its 16-lane group) + 4 q((l % 16) % 4),
with q(u) = 2 (u & 1) + (u >> 1) where the forward image's columns are in chained (phi) order and q(u) = u where they are natural
(W1).  All five backward operands of nerf_mlp_bf16.hip are checked (images as csrc/nerf_mlp_bf16.hip::stage_weights builds them,
W5's image padded to 16 zero rows).  """
import numpy as np

rng = np.random.default_rng(0)
IN, H, X2, PE = 32, 64, 42, 27
ONES = 16 + PE


def phi16(p): return 8 * ((p & 7) >> 2) + 4 * (p >> 3) + (p & 3)
def phi(s): return (s & ~15) + phi16(s & 15)


W1 = rng.normal(size=(H, IN)); W2 = rng.normal(size=(16, H)); W3 = rng.normal(size=(H, X2)); W4 = rng.normal(size=(H, H)); W5 = rng.normal(size=(3, H))
# forward images [out row][K slot]
W1p = W1.copy()
W2p = np.array([[W2[r, phi(s)] for s in range(64)] for r in range(16)])
W3p = np.zeros((64, 48))
for r in range(64):
    for s in range(16):
        m = phi16(s)
        W3p[r, s] = W3[r, m - 1] if m else 0.0
W4p = np.array([[W4[r, phi(s)] for s in range(64)] for r in range(64)])
W5p = np.zeros((16, 64)); W5p[:3] = [[W5[r, phi(s)] for s in range(64)] for r in range(3)]
# transposed images the backward reads today [in row][out-neuron slot]
W5T = np.array([[W5[p, k] if p < 3 else 0.0 for p in range(16)] for k in range(64)])
W4T = np.array([[W4[phi(s), k] for s in range(64)] for k in range(64)])
W3T = np.array([[W3[phi(s), m - 1] if m else 0.0 for s in range(64)] for m in range(16)])
W2T = np.array([[W2[phi16(p), k] for p in range(16)] for k in range(64)])
W1T = np.array([[W1[phi(s), k] for s in range(64)] for k in range(32)])


def tr_read(img, addr):
    out = np.zeros((64, 4))
    for l in range(64):
        i = l % 16
        for j in range(4):
            r, c = addr[16 * (l // 16) + 4 * j + i // 4]
            out[l, j] = img[r, c + i % 4]
    return out


def q_chained(u): return 2 * (u & 1) + (u >> 1)
def q_natural(u): return u


def operand_from_forward(img, kb, col_block_of_group, q, natural_rows=False):
    """col_block_of_group(grp) -> first column of the 16 columns the lanes 16 grp .. 16 grp + 15 (rows m) stand for.
    natural_rows: the K slots of this operand are in natural order (W5T: slot p = colour channel p) instead of chained."""
    got = np.zeros((64, 8))
    for half in range(2):
        addr = []
        for l in range(64):
            c, g = l % 16, l >> 5
            row = (8 * g + 4 * half + c // 4) if natural_rows else (4 * g + 8 * half + c // 4)
            addr.append((16 * kb + row, col_block_of_group((l >> 4) & 3) + 4 * q(c % 4)))
        got[:, 4 * half:4 * half + 4] = tr_read(img, addr)
    return got


checks = 0
# dH3 = W5^T dY5 (2 blocks t of 32 h3 neurons; K = 16 natural channel slots)
for t in range(2):
    want = np.array([W5T[32 * t + (l & 31), 8 * (l >> 5):8 * (l >> 5) + 8] for l in range(64)])
    got = operand_from_forward(W5p, 0, lambda grp: 32 * t + 16 * (grp & 1), q_chained, natural_rows=True)
    assert np.array_equal(got, want); checks += 1
# dH2 = W4^T dH3
for t in range(2):
    for kb in range(4):
        want = np.array([W4T[32 * t + (l & 31), 16 * kb + 8 * (l >> 5):16 * kb + 8 * (l >> 5) + 8] for l in range(64)])
        got = operand_from_forward(W4p, kb, lambda grp: 32 * t + 16 * (grp & 1), q_chained)
        assert np.array_equal(got, want); checks += 1
# dY2 = W3^T dH2 (16 rows m = n & 15: both 16-lane groups of a half-wave stand for the same columns 0..15)
for kb in range(4):
    want = np.array([W3T[(l & 31) & 15, 16 * kb + 8 * (l >> 5):16 * kb + 8 * (l >> 5) + 8] for l in range(64)])
    got = operand_from_forward(W3p, kb, lambda grp: 0, q_chained)
    assert np.array_equal(got, want); checks += 1
# dH1 = W2^T dY2 (K = 16 chained slots)
for t in range(2):
    want = np.array([W2T[32 * t + (l & 31), 8 * (l >> 5):8 * (l >> 5) + 8] for l in range(64)])
    got = operand_from_forward(W2p, 0, lambda grp: 32 * t + 16 * (grp & 1), q_chained)
    assert np.array_equal(got, want); checks += 1
# dX0 = W1^T dH1 (32 input features, natural columns)
for kb in range(4):
    want = np.array([W1T[l & 31, 16 * kb + 8 * (l >> 5):16 * kb + 8 * (l >> 5) + 8] for l in range(64)])
    got = operand_from_forward(W1p, kb, lambda grp: 16 * (grp & 1), q_natural)
    assert np.array_equal(got, want); checks += 1
print(f"all {checks} backward operand blocks come out of the forward images bit for bit")
