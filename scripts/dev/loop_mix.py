"""Instruction mix of every innermost loop of a kernel's gfx950 assembly (hipcc --cuda-device-only -S): spots register copies
(v_mov) that hipcc leaves inside hot loops.  usage: loop_mix.py file.s [kernel-substring]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
kern = None
i = 0
while i < len(lines):
    ln = lines[i]
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kern = m.group(1)
    if "Inner Loop Header" in ln and kern and want in kern:
        # the label is on this line or the closest label line above
        j = i
        while j > 0 and not re.match(r"^\.LBB\d+_\d+:", lines[j]):
            j -= 1
        label = lines[j].split(":")[0]
        k = i + 1
        body = []
        while k < len(lines) and not re.search(r"s_c?branch\w* " + re.escape(label) + r"\b", lines[k]):
            if re.match(r"^_Z\w+:", lines[k]):
                break
            t = lines[k].strip()
            if t and not t.startswith(";") and not t.startswith("."):
                body.append(t.split()[0])
            k += 1
        c = collections.Counter(body)
        tot = len(body)
        top = ", ".join(f"{a}:{b}" for a, b in c.most_common(7))
        print(f"{kern[:70]} {label} n={tot} mov={c['v_mov_b64_e32'] + c['v_mov_b32_e32']} mfma={sum(v for a, v in c.items() if 'mfma' in a)} | {top}")
    i += 1
