"""What hipcc did to a kernel, from its gfx950 assembly (hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S file.hip -o file.s):
per kernel -- registers, scratch bytes, scratch loads / stores by loop depth, LDS / global loads that sit directly in front of their own
full wait (an exposed round trip each), barriers, MFMAs.  Found this round: forty hoisted address / shuffle-index values spilled by
gin_resident_kernel and reloaded inside its MLP steps (behind vmcnt(0), i.e. behind the chunk DMA), and its fourteen serialized
own-row reads.   usage: isa_lint.py file.s [kernel-name substring]"""
import re, sys
path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path, errors="replace").read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for si in starts:
    name = lines[si].split(":")[0]
    if want not in name:
        continue
    try:
        end = next(i for i in range(si, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    except StopIteration:
        continue
    body = lines[si:end]
    if not any("s_endpgm" in l for l in body):
        continue
    meta = {}
    for l in body:
        m = re.search(r"\.amdhsa_(next_free_vgpr|private_segment_fixed_size|group_segment_fixed_size|accum_offset)\s+(\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    ins = []
    depth = 0
    for l in body:
        m = re.search(r"Depth=(\d+)", l)
        if l.startswith(".LBB"):
            depth = int(m.group(1)) if m else 0
        t = l.strip()
        if l.startswith("\t") and t and not t.startswith((".", ";")):
            ins.append((depth, t))
    sc = {}
    for d, t in ins:
        if t.startswith("scratch_"):
            k = ("load" if "load" in t else "store", d)
            sc[k] = sc.get(k, 0) + 1
    ser_lds = ser_glb = 0
    for (d0, a), (d1, b) in zip(ins, ins[1:]):
        if b.startswith("s_waitcnt"):
            if a.startswith("ds_read") and "lgkmcnt(0)" in b:
                ser_lds += 1
            if a.startswith(("global_load", "buffer_load")) and "vmcnt(0)" in b and "lds" not in a:
                ser_glb += 1
    short = re.sub(r"^_ZN?\d*", "", name)[:70]
    print(f"{short}: vgpr {meta.get('next_free_vgpr')} lds {meta.get('group_segment_fixed_size')} scratch {meta.get('private_segment_fixed_size')} B | "
          f"scratch ops by (kind, loop depth) {dict(sorted(sc.items()))} | ds_read->wait0 {ser_lds} | gload->wait0 {ser_glb} | "
          f"barriers {sum(t.startswith('s_barrier') for _, t in ins)} mfma {sum('v_mfma' in t for _, t in ins)} instr {len(ins)}")

# (second pass, with a kernel name given: global loads whose wait follows within three instructions, by loop depth)
if want:
    for si in starts:
        name = lines[si].split(":")[0]
        if want not in name:
            continue
        try:
            end = next(i for i in range(si, len(lines)) if ".end_amdhsa_kernel" in lines[i])
        except StopIteration:
            continue
        body = lines[si:end]
        ins, depth = [], 0
        for l in body:
            m = re.search(r"Depth=(\d+)", l)
            if l.startswith(".LBB"):
                depth = int(m.group(1)) if m else 0
            t = l.strip()
            if l.startswith("\t") and t and not t.startswith((".", ";")):
                ins.append((depth, t))
        for n in range(len(ins) - 4):
            d, a = ins[n]
            if a.startswith(("global_load", "buffer_load")) and "lds" not in a:
                for m in range(1, 4):
                    if ins[n + m][1].startswith("s_waitcnt") and "vmcnt" in ins[n + m][1]:
                        print(f"   depth {d}: {a[:48]} -> {ins[n + m][1]}")
                        break
