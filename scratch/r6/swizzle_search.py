"""Bank-conflict brute force for ROW-MAJOR K / V tiles read by the 32x32x16 forward (csrc/attention_m32.hip, CHADA_M32_RM):
  K fragment:  ds_read_b128, lane l -> row (l & 31), 16-byte chunk 2 ks + (l >> 5); the hardware serves the lane groups {0-3, 12-15, 20-23, 24-27} and
               {4-7, 8-11, 16-19, 28-31} (+32) together (measured: scratch/ldsbank) -- 16 lanes must hit 16 distinct 16-byte slots of the 256-byte bank row;
  V^T fragment: ds_read_b64_tr_b16 in 32-lane phases, lane (g, ii) -> key row kp * 16 + 4 (g >> 1) + (ii >> 2) (+ 8), chunk 4 db + 2 (g & 1) + ((ii & 3) >> 1),
               8 bytes at (ii & 1) * 8 -- 64 distinct banks per phase.
Checks a swizzle chunk' = chunk ^ f(row) for both, and searches the GF(2)-linear f for dh 192 (rows of 384 bytes: attention.hip's dkv_swz<192>,
derived for 16-row reads, conflicts here).  Result used: dh 96 -> dkv_swz<96>; dh 192 -> f = row bits (2, 3, 1) on chunk bits (0, 1, 2)."""
import itertools


def groups():
    g1 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
    g2 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
    return [g1, g2, [x + 32 for x in g1], [x + 32 for x in g2]]


def check(DH, swz):
    KS, DB = DH // 16, DH // 32
    KVT = 64 if DH == 96 else 32
    for kb in range(KVT // 32):
        for ks in range(KS):
            for grp in groups():
                slots = set()
                for l in grp:
                    row = kb * 32 + (l & 31)
                    ch = (2 * ks + (l >> 5)) ^ swz(row)
                    slots.add(((row * DH * 2 + ch * 16) % 256) // 16)
                if len(slots) != 16:
                    return False
    for kp in range(KVT // 16):
        for db in range(DB):
            for second in (0, 1):
                for half in (0, 1):
                    banks = set()
                    for l in range(32 * half, 32 * half + 32):
                        g, ii = l >> 4, l & 15
                        trow = kp * 16 + 4 * (g >> 1) + (ii >> 2) + 8 * second
                        ch = (4 * db + 2 * (g & 1) + ((ii & 3) >> 1)) ^ swz(trow)
                        addr = trow * DH * 2 + ch * 16 + (ii & 1) * 8
                        for b in range(2):
                            bank = ((addr // 4) + b) % 64
                            if bank in banks:
                                return False
                            banks.add(bank)
    return True


if __name__ == "__main__":
    print("dh 96, dkv_swz<96>:", check(96, lambda r: (4 - ((r >> 2) & 3)) & 3))
    print("dh 192, dkv_swz<192> (row & 6):", check(192, lambda r: r & 6))
    print("dh 192, row bits (2, 3, 1):", check(192, lambda r: ((r >> 2) & 3) | (((r >> 1) & 1) << 2)))
    n = 0
    for m in itertools.product([1, 2, 4, 8, 16], repeat=3):
        f = lambda r, m=m: sum(((r & m[b]) != 0) << b for b in range(3))
        if check(192, f):
            n += 1
            print("  single-bit map", m)
    print(n, "single-bit maps are conflict-free at dh 192")
