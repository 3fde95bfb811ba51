"""The one ordering rule of the LDS-DMA idiom, checked on the gfx950 assembly hipcc produced.

Every weight / row / descriptor stream of the resident and fused kernels is `global_load_lds_dwordx4` (lds_dma16, device_common.h)
-> `s_waitcnt vmcnt(M)` by the ISSUING wave -> `s_barrier` -> `ds_read` by anyone.  hipcc does not know that the inline-asm request
writes LDS, nor that it counts in vmcnt: nothing but the hand-placed wait orders the consumers behind the transfer
(MI355X_MICROARCH.md: "a read issued earlier returns the OLD LDS bytes, no stall").  The rule:

    in program order, between a global_load_lds and the next s_barrier, the issuing wave executes an `s_waitcnt vmcnt(M)`
    with M <= K, K = the vector-memory instructions issued after that request (vmcnt retires in order: the request is complete
    once at most K operations are outstanding) -- vmcnt(0) always qualifies.

A request that reaches a barrier uncovered is reported as CARRIED (legal only where the kernel says so: a later wait + barrier pair
must cover it before the buffer is read -- the tool follows it to that pair and reports how far it was carried).  A request covered
by a wait with M > 0 is reported as COUNTED with (M, K): these depend on how many loads hipcc placed behind the request (the advisor's
round-4 finding on gin_resident_kernel's `vmcnt(8)` / `vmcnt(6)`), so tests/test_dma_lint.py pins their number per kernel and requires
the snapshot to be regenerated and re-read when it moves (a wait that stops covering shows up as a later cover or a carry).  The walk
follows every control-flow path from the request (K = the minimum over paths; paths hipcc's layout allows but the kernel's
conditions exclude -- "no next tile" after a next-tile request -- show up as requests still in flight at s_endpgm, reported, harmless).

usage: dma_lint.py file.s [kernel-name substring]      (hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S file.hip -o file.s)"""
import re
import sys

VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_load", "scratch_store",
        "flat_load", "flat_store", "flat_atomic")


def kernels(path):
    """-> (name, ins, labels): ins = instruction texts in layout order, labels = {label: index of the instruction it precedes}"""
    lines = open(path, errors="replace").read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si in starts:
        name = lines[si].split(":")[0]
        try:
            end = next(i for i in range(si, len(lines)) if ".end_amdhsa_kernel" in lines[i])
        except StopIteration:
            continue
        body = lines[si:end]
        if not any("s_endpgm" in l for l in body):
            continue
        ins, labels = [], {}
        for l in body:
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                labels[m.group(1)] = len(ins)
                continue
            t = l.strip()
            if l.startswith("\t") and t and not t.startswith((".", ";")):
                ins.append(t.split(";")[0].strip())
        yield name, ins, labels


def vmcnt_of(t):
    """M of an s_waitcnt that names vmcnt, None if it does not constrain vmcnt."""
    if not t.startswith("s_waitcnt"):
        return None
    m = re.search(r"vmcnt\((\d+)\)", t)
    if m:
        return int(m.group(1))
    m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", t)  # raw immediate: gfx9 vmcnt = bits [3:0] | [15:14] << 4
    if m:
        v = int(m.group(1), 0)
        return (v & 0xF) | (((v >> 14) & 0x3) << 4)
    return None


def lint(ins, labels):
    """Every control-flow path from every request, until a covering wait (per path: K = the vector-memory instructions on it).
    -> dict(requests, covered0, counted=[(M, min K)], carried=[barriers crossed before the cover], uncovered)"""
    res = {"requests": 0, "covered0": 0, "counted": [], "carried": [], "uncovered": 0}
    n = len(ins)
    for i, t in enumerate(ins):
        if not t.startswith("global_load_lds"):
            continue
        res["requests"] += 1
        # depth-first over (pc, K, barriers crossed); a pc revisited with no smaller K and no more barriers adds nothing
        best = {}
        stack = [(i + 1, 0, 0)]
        covers = {}  # wait pc -> (M, min K, max barriers)
        uncovered = False
        while stack:
            pc, k, bars = stack.pop()
            while True:
                if pc >= n:
                    uncovered = True
                    break
                seen = best.get(pc)
                if seen is not None and seen[0] <= k and seen[1] >= bars:
                    break
                best[pc] = (k if seen is None else min(k, seen[0]), bars if seen is None else max(bars, seen[1]))
                u = ins[pc]
                if u.startswith(VMEM):
                    k += 1
                else:
                    m = vmcnt_of(u)
                    if m is not None and m <= k:
                        c = covers.get(pc)
                        covers[pc] = (m, k if c is None else min(k, c[1]), bars if c is None else max(bars, c[2]))
                        break
                    if u.startswith("s_barrier"):
                        bars += 1
                    elif u.startswith("s_endpgm"):
                        uncovered = True  # the wave ends with the request in flight (statically possible paths only; harmless for LDS)
                        break
                    elif u.startswith("s_branch"):
                        pc = labels[u.split()[-1]]
                        continue
                    elif u.startswith("s_cbranch"):
                        stack.append((labels[u.split()[-1]], k, bars))
                pc += 1
        if uncovered:
            res["uncovered"] += 1
        for m, k, bars in covers.values():
            if m == 0:
                res["covered0"] += 1
            else:
                res["counted"].append((m, k))
            if bars:
                res["carried"].append(bars)
    return res


def snapshot(path):
    """{kernel: summary} of every kernel of an assembly file that issues LDS-DMA (what tests/test_dma_lint.py pins)"""
    out = {}
    for name, ins, labels in kernels(path):
        r = lint(ins, labels)
        if r["requests"]:
            out[name] = {"requests": r["requests"], "covers_vmcnt0": r["covered0"], "counted": sorted(set(map(tuple, r["counted"]))),
                         "carried": sorted(r["carried"]), "in_flight_at_end": r["uncovered"]}
    return out


def main():
    if sys.argv[1] == "--json":
        import json
        print(json.dumps({p.split("/")[-1]: snapshot(p) for p in sys.argv[2:]}, indent=1, sort_keys=True))
        return
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, ins, labels in kernels(path):
        if want not in name:
            continue
        r = lint(ins, labels)
        if not r["requests"]:
            continue
        short = re.sub(r"^_ZN?\d*", "", name)[:80]
        print(f"{short}: {r['requests']} LDS-DMA requests | covering waits vmcnt(0): {r['covered0']} | counted (M, K): {sorted(set(r['counted']))} x{len(r['counted'])} | "
              f"carried across barriers: {len(r['carried'])} (max {max(r['carried']) if r['carried'] else 0}) | in flight at s_endpgm on some static path: {r['uncovered']}")


if __name__ == "__main__":
    main()
