"""Audit of a kernel .s listing (hipcc -S --cuda-device-only, one kernel cut out with awk) for hipcc touching a\nregister between the inline-asm global_load that fills it and the s_waitcnt that retires it (straight-line\napproximation: a fully unrolled kernel).  python asm_load_audit.py kernel.s"""
import re, sys
lines=open(sys.argv[1]).read().split('\n')
pending={}  # reg -> line index of its load
vmops=[]
viol=[]
def regs_in(tok):
    out=set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2))+1))
    tok2=re.sub(r'v\[\d+:\d+\]','',tok)
    for m in re.finditer(r'\bv(\d+)\b', tok2):
        out.add(int(m.group(1)))
    return out
for n,l in enumerate(lines):
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    if t.startswith('global_load_dwordx4'):
        m=re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\], (v\[\d+:\d+\]|v\d+)', t)
        for r in regs_in(m.group(3)):
            if r in pending: viol.append((n,t,'addr uses pending reg v%d'%r))
        vmops.append(n)
        for r in range(int(m.group(1)), int(m.group(2))+1):
            pending[r]=n
        continue
    if t.startswith('global_load_lds') or t.startswith('global_load_dword ') or t.startswith('global_load_dwordx2'):
        vmops.append(n)
    if t.startswith('s_waitcnt') and 'vmcnt' in t:
        cnt=int(re.search(r'vmcnt\((\d+)\)',t).group(1))
        keep=set(vmops[-cnt:]) if cnt>0 else set()
        pending={r:ln for r,ln in pending.items() if ln in keep}
        continue
    used=regs_in(t)
    bad=[r for r in used if r in pending]
    if bad: viol.append((n,t,'touches pending '+','.join('v%d'%r for r in bad[:6])))
print(len(viol),'violations (straight-line approximation)')
for v in viol[:25]: print(v)
