"""Conflict search for the split (three bf16 planes) deep kernels: pixels of 16 bytes (8 channels), ds_read_b128 lane groups."""
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
          [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59], [36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
def cost_down(H, W, NS, WP, PLANE):
    Hs, Ws = H // 2 + 1, W // 2 + 1
    P = Hs * Ws; N = NS * P; NT = (N + 15) // 16
    tot = 0
    for t in range(NT):
        for kh in range(4):
            for grp in GROUPS:
                slots = {}
                for lane in grp:
                    n = min(t * 16 + (lane & 15), N - 1); kq = lane >> 4
                    si, pix = divmod(n, P); oh, ow = divmod(pix, Ws)
                    px = si * PLANE + (2 * oh + kh) * WP + 2 * ow + kq
                    slots.setdefault(px % 16, set()).add(px)
                tot += max(len(v) for v in slots.values()) - 1
    return tot, NT * 4 * 4
for (H, W, NS) in [(5, 7, 8), (5, 7, 4), (9, 12, 4), (9, 12, 2), (17, 23, 1), (17, 23, 2)]:
    Hs, Ws = H // 2 + 1, W // 2 + 1
    HP = 2 * Hs + 2
    best = []
    for WP in range(2 * Ws + 2, 2 * Ws + 20):
        for PLANE in range(HP * WP, HP * WP + 24):
            c, n = cost_down(H, W, NS, WP, PLANE)
            best.append((c, PLANE, WP, n))
    best.sort()
    print((H, W, NS), 'HP', HP, 'min WP', 2 * Ws + 2, best[:4])

print('--- up: 16-byte pixels, waves = (phase, M half), tiles of the phase')
def cost_up(H, W, NS, SWP, SPLANE):
    Hs, Ws = H // 2 + 1, W // 2 + 1
    tot = cnt = 0
    for ph in range(4):
        hu = H // 2 if ph >> 1 else (H + 1) // 2
        wu = W // 2 if ph & 1 else (W + 1) // 2
        npx = NS * hu * wu
        for t in range((npx + 15) // 16):
            for grp in GROUPS:
                slots = {}
                for lane in grp:
                    n = min(t * 16 + (lane & 15), npx - 1); kq = lane >> 4; th, tw = kq >> 1, kq & 1
                    si, rem = divmod(n, hu * wu); u, v = divmod(rem, wu)
                    px = si * SPLANE + (u + 1 - th) * SWP + v + 1 - tw
                    slots.setdefault(px % 16, set()).add(px)
                tot += max(len(v) for v in slots.values()) - 1
                cnt += 1
    return tot, cnt
for (H, W, NS) in [(5, 7, 8), (9, 12, 4), (17, 23, 2), (17, 23, 1)]:
    Hs, Ws = H // 2 + 1, W // 2 + 1
    best = []
    for SWP in range(Ws + 1, Ws + 14):
        for SPLANE in range((Hs + 1) * SWP, (Hs + 1) * SWP + 20):
            c, n = cost_up(H, W, NS, SWP, SPLANE)
            best.append((c, SPLANE, SWP, n))
    best.sort()
    print((H, W, NS), 'rows', Hs + 1, 'min SWP', Ws + 1, best[:4])
