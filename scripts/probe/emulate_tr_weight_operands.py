"""Can the decoder backward read its transposed weight operands out of the FORWARD weight images (dropping the 22 KB of
transposed copies in LDS)?  Index-logic check in float64, with the ds_read_b64_tr_b16 lane / element mapping that
scripts/probe/probe_mfma.hip confirmed on gfx950: result element j of lane i of a 16-lane group = element i % 4 of the 8-byte chunk
addressed by lane 4 j + i / 4 of that group.

A operand of dH = W4^T dY (v_mfma_f32_32x32x16_bf16, lane l: row m = l & 31, K slots 8 g .. 8 g + 7 of block kb, g = l >> 5),
input-neuron block t: today `W4T[32 t + m][16 kb + 8 g + 0..7]` (one ds_read_b128 from the transposed image); here two transposing
reads from the forward image `W4p[row][slot] = W4[row][phi(slot)]`, lane l supplying the chunk address
    row = 16 kb + 4 g (+ 8 for the second read) + (l % 16) / 4,   column = 32 t + 16 ((l >> 4) & 1) + 4 q,
    q = 2 (((l % 16) % 4) & 1) + (((l % 16) % 4) >> 1).
"""
import numpy as np

rng = np.random.default_rng(0)
H = 64


def phi16(p): return 8 * ((p & 7) >> 2) + 4 * (p >> 3) + (p & 3)
def phi(s): return (s & ~15) + phi16(s & 15)


W4 = rng.normal(size=(H, H))
W4p = np.array([[W4[r, phi(s)] for s in range(H)] for r in range(H)])       # forward image
W4T = np.array([[W4[phi(s), k] for s in range(H)] for k in range(H)])       # transposed image the backward reads today


def tr_read(addr):
    out = np.zeros((64, 4))
    for l in range(64):
        i = l % 16
        for j in range(4):
            r, c = addr[16 * (l // 16) + 4 * j + i // 4]
            out[l, j] = W4p[r, c + i % 4]
    return out


bad = 0
for t in range(2):
    for kb in range(4):
        want = np.array([W4T[32 * t + (l & 31), 16 * kb + 8 * (l >> 5):16 * kb + 8 * (l >> 5) + 8] for l in range(64)])
        got = np.zeros((64, 8))
        for half in range(2):
            addr = []
            for l in range(64):
                c, g = l % 16, l >> 5
                q = 2 * ((c % 4) & 1) + ((c % 4) >> 1)
                addr.append((16 * kb + 4 * g + 8 * half + c // 4, 32 * t + 16 * ((l >> 4) & 1) + 4 * q))
            got[:, 4 * half:4 * half + 4] = tr_read(addr)
        bad += int(np.abs(got - want).max() > 0)
print("operand blocks that differ from the transposed image's:", bad, "of 8")
assert bad == 0
