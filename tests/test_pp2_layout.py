"""CPU model of gemm_pp2_kernel's LDS stage (csrc/gemm_impl.h, round 4): the two-plane 128x128 tile with the ping-pong schedule.
Stage image (rows of 128 bytes = one 64-element k-tile row): [A0 hi 64 | A0 lo 64 | A1 hi 64 | A1 lo 64 | W hi 128 | W lo 128];
wave group 0 (rows 0..63 of the tile) loads W and group 1's A rows, group 1 loads group 0's A rows.  The test replays the
loader's and the reader's address arithmetic: every (plane, tile row, k chunk) a wave's MFMA fragment read asks for must be the
one some lane's LDS-DMA put there, each LDS position is written exactly once per k-tile, and the pieces split 12 : 4 per wave."""
import numpy as np

Q = 64 * 128            # bytes of a 64-row operand tile
STAGE = 8 * Q


def a_ptr(g):           # [hi 64 rows | lo 64 rows] of group g
    return g * 2 * Q


W_PTR = 4 * Q


def loader_writes():
    """-> dict: LDS byte offset (16-byte granularity) -> ('A' | 'W', plane, source row in the tile, source chunk), and the
    number of DMA instructions per wave of each group."""
    img, pieces = {}, {0: 0, 1: 0}
    for tid in range(512):
        wave, lane = tid >> 6, tid & 63
        wm, wq = wave >> 2, wave & 3
        t = tid & 255
        kc, r0 = t & 7, t >> 3
        sc = kc ^ ((r0 >> 1) & 7)
        a_row0 = 64 if wm == 0 else 0          # group 0 loads the A rows of group 1 and vice versa

        def put(dst_tile, pass_row, kind, plane, src_row):
            # the DMA image is lane-linear: lane l of a wave writes 16 bytes at  base + wq*1024 + pass_row*128 ... + l*16
            off = dst_tile + wq * 1024 + pass_row * 128 + lane * 16
            assert off not in img, (off, img.get(off))
            img[off] = (kind, plane, src_row, sc)
            # ... which is row (lane >> 3) of the wave's 8 rows, chunk lane & 7 -- the thread's own (r0, kc)
            assert (wq * 8 + (lane >> 3)) == r0 % 32 and (lane & 7) == kc

        for i in range(2):                      # this group's A pieces: hi and lo plane of rows a_row0 + r0 + 32 i
            put(a_ptr(wm ^ 1), 32 * i, "A", 0, a_row0 + r0 + 32 * i)
            put(a_ptr(wm ^ 1) + Q, 32 * i, "A", 1, a_row0 + r0 + 32 * i)
        if lane == 0:
            pieces[wm] += 4
        if wm == 0:                             # group 0 also loads W: passes 0..3 hi, 4..7 lo
            for j in range(8):
                put(W_PTR, 32 * j, "W", j >> 2, r0 + 32 * (j & 3))
            if lane == 0:
                pieces[0] += 8
    return img, pieces


def test_every_fragment_read_finds_its_operand():
    img, pieces = loader_writes()
    assert len(img) == STAGE // 16              # every 16-byte position of the stage written exactly once
    assert pieces == {0: 4 * 12, 1: 4 * 4}      # 12 pieces per wave in group 0, 4 in group 1
    for wave in range(8):
        wm, wn = wave >> 2, wave & 3
        for lane in range(64):
            lr, lh = lane & 31, lane >> 5
            for ks in range(4):
                chunk = 2 * ks + lh
                for i in range(2):              # A fragments: mma_tile<.., TM = 2, TN = 1>(sa = a_ptr(wm), a_lo = Q, wm = 0)
                    row = i * 32 + lr
                    off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
                    for plane in range(2):
                        kind, pl, src_row, src_chunk = img[a_ptr(wm) + plane * Q + off]
                        assert (kind, pl, src_row, src_chunk) == ("A", plane, 64 * wm + row, chunk)
                row = wn * 32 + lr              # W fragment: sb = W_PTR, b_lo = 128 * 128
                off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
                for plane in range(2):
                    kind, pl, src_row, src_chunk = img[W_PTR + plane * 128 * 128 + off]
                    assert (kind, pl, src_row, src_chunk) == ("W", plane, row, chunk)
