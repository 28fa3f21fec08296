"""LDS strides of conv_wgrad_split.hip: extra ds_read_b128 cycles (bank conflicts inside the instruction's four 16-lane groups)
of the A (gradient / small operand) and B (parity-split big operand) fragment reads, per candidate channel strides."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def extra_cycles(addr_of_lane):
    """addr in bf16 elements (16-byte aligned fragments); returns extra LDS cycles of one ds_read_b128"""
    tot = 0
    for g in GROUPS:
        slots = {}
        for l in g:
            a = addr_of_lane(l) // 8           # 16-byte slot
            slots.setdefault(a % 16, set()).add(a)
        tot += max(len(v) for v in slots.values()) - 1
    return tot


def search(CB, CS, Ws, R):
    GPR = (Ws + 7) // 8                      # 8-pixel groups per row
    SROW = GPR * 8                         # X rows
    SROWS = GPR * 8 + 8                    # S rows: + 8 zeros (the shifted fragment of the last group reads one dword on)
    XR = 2 * R + 2
    ng = R * GPR
    nks = (ng + 3) // 4
    best = None
    for sch in range(R * SROWS, R * SROWS + 136, 8):
        c = 0
        for ks in range(nks):
            def a(l, ks=ks):
                m, kq = l & 15, l >> 4
                gi = min(4 * ks + kq, ng - 1)
                return m * sch + (gi // GPR) * SROWS + (gi % GPR) * 8
            c += extra_cycles(a)
        if best is None or c < best[0]:
            best = (c, sch)
    bestx = None
    for xch in range(XR * SROW, XR * SROW + 136, 8):
        for xplpad in range(0, 136, 8):
            xpl = CB * xch + xplpad
            c = 0
            for ks in range(nks):
                for kh_dummy in (0,):
                    def b(l, ks=ks):
                        n, kq = l & 15, l >> 4
                        gi = min(4 * ks + kq, ng - 1)
                        return (n & 1) * xpl + (n >> 3) * xch + (2 * (gi // GPR) + ((n >> 1) & 3)) * SROW + (gi % GPR) * 8
                    c += extra_cycles(b)
            if bestx is None or c < bestx[0]:
                bestx = (c, xch, xpl)
    print(f'CB {CB} CS {CS} Ws {Ws} R {R}: groups/row {GPR} K steps {nks}; A: SCH {best[1]} extra {best[0]} / {nks} reads; '
          f'B: XCH {bestx[1]} XPL {bestx[2]} extra {bestx[0]} / {nks} reads; LDS {3 * 2 * (CS * best[1] + 2 * bestx[2]) / 1024:.1f} KB')


for CB, CS, Ws, R in ((8, 16, 88, 4), (16, 32, 45, 4), (32, 64, 23, 4), (8, 16, 88, 2), (16, 32, 45, 2), (32, 64, 23, 2)):
    search(CB, CS, Ws, R)
