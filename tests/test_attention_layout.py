"""CPU model of the attention kernel's V path (csrc/attention.hip, round 4): the V tile travels row-major by LDS-DMA with a
chunk swizzle and is transposed by the fragment read.  The semantics of ds_read_b64_tr_b16 were MEASURED (tools/gpu/att_probe.py,
profiles/r04_experiments.md 6): within a 16-lane group lane i receives element i % 4 of the 8-byte chunks whose addresses
lanes i / 4, 4 + i / 4, 8 + i / 4, 12 + i / 4 supplied.  This test replays the kernel's address arithmetic against that model and
checks what the PV MFMA needs: lane (lr, lh) of A-fragment (dt, i) holds V[key][d = 32 dt + lr] for the eight keys
16 i + (e & 3) + 8 (e >> 2) + 4 lh -- the key order of the P^T accumulator registers -- and that one transposing read is free
of avoidable bank conflicts (its eight rows fill two whole 256-byte bank rows)."""
import numpy as np

KT, D = 64, 64  # keys per tile, head dimension


def lds_image_of_v_tile():
    """What dma_kv leaves in LDS: piece p = 8 rows; lane l of the issuing wave writes chunk l & 7 of row 8 p + (l >> 3) and
    fetches source chunk (l & 7) ^ (4 * ((row >> 1) & 1)).  Element value = key * 64 + d."""
    img = np.full(KT * D, -1, dtype=np.int64)  # 16-bit elements, row stride 64 elements (128 bytes)
    for piece in range(8):
        for lane in range(64):
            row = piece * 8 + (lane >> 3)
            src_chunk = (lane & 7) ^ (4 * ((row >> 1) & 1))
            dst = row * 64 + (lane & 7) * 8
            img[dst:dst + 8] = row * 64 + src_chunk * 8 + np.arange(8)
    assert (img >= 0).all()
    return img


def tr_read(img, byte_addr):
    """ds_read_b64_tr_b16 of one wave: byte_addr[64] -> out[64][4] (the measured semantics)."""
    out = np.zeros((64, 4), dtype=np.int64)
    for lane in range(64):
        g, i = lane // 16, lane % 16
        for j in range(4):
            src_lane = g * 16 + 4 * j + i // 4
            a = byte_addr[src_lane]
            assert a % 8 == 0
            out[lane, j] = img[a // 2 + i % 4]
    return out


def kernel_addresses(dt, i, half):
    """va[dt] + i * 2048 (+ 1024 for the second half of the eight keys), as in attention.hip."""
    addr = np.zeros(64, dtype=np.int64)
    for lane in range(64):
        lh, lq, g16 = lane >> 5, lane & 15, (lane >> 4) & 1
        va = (4 * lh + (lq >> 2)) * 128 + (((dt ^ ((lq >> 3) & 1)) * 4 + 2 * g16 + ((lq & 3) >> 1)) << 4) + (lq & 1) * 8
        addr[lane] = va + i * 2048 + half * 1024
    return addr


def test_transposing_read_delivers_the_mfma_fragment():
    img = lds_image_of_v_tile()
    for dt in range(2):
        for i in range(4):
            frag = np.concatenate([tr_read(img, kernel_addresses(dt, i, 0)), tr_read(img, kernel_addresses(dt, i, 1))], axis=1)  # [64][8]
            for lane in range(64):
                lr, lh = lane & 31, lane >> 5
                for e in range(8):
                    key = 16 * i + (e & 3) + 8 * (e >> 2) + 4 * lh
                    assert frag[lane, e] == key * 64 + 32 * dt + lr, (dt, i, lane, e)


def test_one_transposing_read_fills_two_bank_rows():
    """Banks are 4 bytes wide, 64 of them (256 bytes) per LDS clock.  The 64 lanes of one read touch 64 x 8 = 512 bytes: two
    clocks at best.  With the chunk swizzle the 8-byte accesses of a read cover every bank exactly twice."""
    for dt in range(2):
        for half in range(2):
            addr = kernel_addresses(dt, 1, half)
            banks = np.concatenate([(addr // 4) % 64, (addr // 4 + 1) % 64])
            counts = np.bincount(banks, minlength=64)
            assert counts.min() == 2 and counts.max() == 2, counts


def test_k_tile_image_matches_its_fragment_reads():
    """K keeps the GEMM kernels' scheme: position chunk c of row r holds source chunk c ^ ((r >> 1) & 7); the QK^T fragment of
    lane (lr, lh), k-step ks reads chunk (2 ks + lh) ^ ((lr >> 1) & 7) of rows lr and 32 + lr -- it must find d = 16 ks + 8 lh ..."""
    img = np.full(KT * D, -1, dtype=np.int64)
    for piece in range(8):
        for lane in range(64):
            row = piece * 8 + (lane >> 3)
            src_chunk = (lane & 7) ^ ((row >> 1) & 7)
            img[row * 64 + (lane & 7) * 8: row * 64 + (lane & 7) * 8 + 8] = row * 64 + src_chunk * 8 + np.arange(8)
    for lane in range(64):
        lr, lh = lane & 31, lane >> 5
        for ks in range(4):
            chunk = 2 * ks + lh
            off = lr * 128 + ((chunk ^ ((lr >> 1) & 7)) << 4)
            for half in range(2):
                got = img[(off + half * 32 * 128) // 2: (off + half * 32 * 128) // 2 + 8]
                want = (lr + 32 * half) * 64 + 16 * ks + 8 * lh + np.arange(8)
                assert (got == want).all()
