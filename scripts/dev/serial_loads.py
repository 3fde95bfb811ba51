"""Finds global loads that hipcc placed directly in front of their own s_waitcnt vmcnt(0) (an exposed round trip each), per kernel,
in gfx950 assembly (hipcc --cuda-device-only -S).  usage: serial_loads.py file.s"""
import re,sys
lines=open(sys.argv[1]).read().split("\n")
kern=None; out={}
for i,l in enumerate(lines):
    m=re.match(r"^(_Z\w+):",l)
    if m: kern=m.group(1)
    t=l.strip()
    if kern and re.match(r"(global|buffer)_load_(dword|ubyte|ushort|sbyte|short)",t) and "lds" not in t.split()[0]:
        # look ahead up to 4 instrs for vmcnt(0)
        k=i+1;n=0
        while k<len(lines) and n<3:
            u=lines[k].strip()
            if u and not u.startswith(";") and not u.startswith("."):
                n+=1
                if u.startswith("s_waitcnt") and "vmcnt(0)" in u:
                    out.setdefault(kern,[]).append(i+1); break
                if re.match(r"(global|buffer)_load",u): break
            k+=1
for k,v in out.items():
    if len(v)>=3: print(len(v),k[:90],v[:12])
