"""LDS bank-conflict model of the row kernels' workgroup FFT (csrc/lpc_sfft.h) on gfx950.

Hardware rules (MI355X_MICROARCH.md, LDS): an 8-byte access (real2 in the float32 build) is a `ds_read_b64` --
serviced in two 32-lane groups, bank = (byte address / 4) mod 64 -- or a `ds_write_b64` -- four contiguous 16-lane
groups, bank = (byte address / 4) mod 32.  Within a group every extra distinct dword address on a busy bank costs one
more LDS cycle.  A conflict-free read takes 2 array cycles per wave instruction, a write 4 (its issue cost is ~6, so a
write only gets slower once its array cycles exceed 6).

    python tools/lds_model.py 4096 16.16.16 512            # half-length rows of 12 MP
    python tools/lds_model.py 960 8.8.5.3 128 --paired     # C4

prints, per access site of the forward / inverse half-row (or paired-row) kernel, the array cycles per wave
instruction under each candidate layout: `none`, `skew8` (slot = i + i/8, the layout of rounds 1-3) and `xor` variants.
"""
import argparse
import math
from collections import defaultdict


def layouts(n):
    out = {
        "none": lambda i: i,
        "skew8": lambda i: i + (i >> 3),
        "skew16": lambda i: i + (i >> 4),
        "skew32": lambda i: i + (i >> 5),
        "pad32x1": lambda i: i + (i >> 5),
        "xor4": lambda i: i ^ ((i >> 4) & 15),
        "xor5": lambda i: i ^ ((i >> 5) & 31),
        "xor4b": lambda i: i ^ ((i >> 8) & 15) ^ ((i >> 4) & 15),
    }
    return out


def read_cycles(slots):
    """ds_read_b64: lanes in two groups of 32, 64 banks of dwords"""
    total = 0
    for g0 in range(0, len(slots), 32):
        grp = [s for s in slots[g0:g0 + 32] if s is not None]
        if not grp:
            continue
        per_bank = defaultdict(set)
        for s in grp:
            for d in (2 * s, 2 * s + 1):
                per_bank[d % 64].add(d)
        total += max(len(v) for v in per_bank.values())
    return total


def write_cycles(slots):
    """ds_write_b64: four contiguous groups of 16 lanes, 32 banks of dwords"""
    total = 0
    for g0 in range(0, len(slots), 16):
        grp = [s for s in slots[g0:g0 + 16] if s is not None]
        if not grp:
            continue
        per_bank = defaultdict(set)
        for s in grp:
            for d in (2 * s, 2 * s + 1):
                per_bank[d % 32].add(d)
        total += max(len(v) for v in per_bank.values())
    return total


def stage_sites(n, radices, nt, first_fused=True, last_fused=True):
    """(name, kind, [per wave-instruction lists of element indices per lane])"""
    sites = []
    ns = 1
    for st, r in enumerate(radices):
        nb = n // r
        maxb = (nb + nt - 1) // nt
        reads, writes = [], []
        for wave0 in range(0, nt, 64):
            for b in range(maxb):
                for m in range(r):
                    rl, wl = [], []
                    for lane in range(64):
                        w = wave0 + lane + b * nt
                        if w >= nb:
                            rl.append(None)
                            wl.append(None)
                            continue
                        rl.append(w + m * nb)
                        jq, k = divmod(w, ns)
                        wl.append(jq * ns * r + k + m * ns)
                    reads.append(rl)
                    writes.append(wl)
        if not (st == 0 and first_fused):
            sites.append((f"stage{st} r{r} read", "r", reads))
        if not (st == len(radices) - 1 and last_fused):
            sites.append((f"stage{st} r{r} write", "w", writes))
        ns *= r
    return sites


def tangle_sites(n, nt, half):
    """Hermitian (un)tangling: lane k touches k and M - k (half rows: M = n; paired rows: the transform length)"""
    reads = []
    for wave0 in range(0, nt, 64):
        q = 0
        while wave0 + q * nt <= n // 2:
            a, b = [], []
            for lane in range(64):
                k = wave0 + lane + q * nt
                if k > n // 2:
                    a.append(None)
                    b.append(None)
                else:
                    a.append(k)
                    b.append(0 if k == 0 else n - k)
            reads.append(a)
            reads.append(b)
            q += 1
    return [("untangle read (k, M-k)", "r", reads), ("tangle write (k, M-k)", "w", reads)]


def fill_sites(n, nt):
    ins = []
    for wave0 in range(0, nt, 64):
        k = 0
        while wave0 + k * nt < n:
            ins.append([wave0 + lane + k * nt if wave0 + lane + k * nt < n else None for lane in range(64)])
            k += 1
    return [("tile fill / drain (contiguous)", "w", ins), ("tile drain (contiguous)", "r", ins)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("radices")
    ap.add_argument("nt", type=int)
    ap.add_argument("--layouts", default="none,skew8,xor4,xor5")
    args = ap.parse_args()
    rad = [int(v) for v in args.radices.split(".")]
    assert math.prod(rad) == args.n
    lays = layouts(args.n)
    names = args.layouts.split(",")
    sites = stage_sites(args.n, rad, args.nt, first_fused=False, last_fused=False) + tangle_sites(args.n, args.nt, True) \
        + fill_sites(args.n, args.nt)
    print(f"{'site':34s} {'instr':>6s} {'ideal':>6s} " + " ".join(f"{nm:>8s}" for nm in names))
    tot = {nm: 0 for nm in names}
    tot_ideal = 0
    for name, kind, instrs in sites:
        fn = read_cycles if kind == "r" else write_cycles
        ideal = sum(fn(list(range(len([x for x in il if x is not None])))) for il in instrs)
        row = []
        for nm in names:
            lay = lays[nm]
            c = sum(fn([None if x is None else lay(x) for x in il]) for il in instrs)
            row.append(c)
            tot[nm] += c
        tot_ideal += ideal
        print(f"{name:34s} {len(instrs):6d} {ideal:6d} " + " ".join(f"{c:8d}" for c in row))
    print(f"{'total array cycles / workgroup':34s} {'':6s} {tot_ideal:6d} " + " ".join(f"{tot[nm]:8d}" for nm in names))


if __name__ == "__main__":
    main()
