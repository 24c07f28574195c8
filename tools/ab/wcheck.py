import re,sys
lines=open(sys.argv[1]).read().split('\n')
ranges=[tuple(int(x) for x in r.split('-')) for r in sys.argv[2:]]
body=[]
lnos=[]
for a,b in ranges:
    body+=lines[a-1:b]; lnos+=list(range(a,b+1))
def regs(tok):
    out=set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]',tok):
        out.update(range(int(m.group(1)),int(m.group(2))+1))
    for m in re.finditer(r'\bv(\d+)\b',tok):
        out.add(int(m.group(1)))
    return out
pending=[]  # list of (kind, dstregs, text, lineno)
viol=0
for it in range(2):
  for i,l in enumerate(body):
    ln=lnos[i]
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    t=t.split(';')[0].strip()
    op=t.split()[0]
    args=t[len(op):]
    if op=='s_waitcnt':
        m=re.search(r'vmcnt\((\d+)\)',t)
        if m:
            n=int(m.group(1))
            while len(pending)>n: pending.pop(0)
        continue
    isvm=op.startswith(('global_','scratch_','buffer_','flat_'))
    isload=isvm and 'load' in op
    parts=[a.strip() for a in args.split(',')]
    if isload:
        dst=regs(parts[0]); src=set().union(*[regs(p) for p in parts[1:]]) if len(parts)>1 else set()
    elif isvm:
        dst=set(); src=regs(args)
    else:
        # VALU/DS etc: first operand dst (approx), all operands may be read
        dst=regs(parts[0]) if parts else set(); src=regs(args)
    for k,d,txt,pl in pending:
        if k=='load' and ((src|dst)&d):
            if it==1 or True:
                print("VIOLATION line %d: %s   touches pending load dst of line %d: %s  (outstanding %d)"%(ln,t,pl,txt,len(pending)))
                viol+=1
            break
    if isvm:
        pending.append(('load' if isload else 'store',dst,t,ln))
print("violations:",viol)
