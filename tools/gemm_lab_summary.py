import re,collections,sys
rows=collections.OrderedDict()
cur=None
for l in open(sys.argv[1]):
    if l.startswith('=='):
        cur=l.split()[1].rstrip(','); continue
    m=re.match(r'M=(\d+) N=(\d+) K=(\d+) g=(\d): ([\d.]+) ms (\d+) TF/s(?: \| wg0: (\d+) cyc/K-tile, epilogue (\d+), total (\d+)/tile, ([\d.]+) GHz)?',l)
    if m:
        key=(m.group(1),m.group(2),m.group(3),m.group(4))
        rows.setdefault(cur,collections.OrderedDict()).setdefault(key,[]).append((float(m.group(5)),int(m.group(6)),m.group(7),m.group(8),m.group(10)))
shapes=list(next(iter(rows.values())).keys())
print("variant".ljust(24)+" ".join(("%sx%sx%s%s"%(s[0][:3],s[1],s[2],'g' if s[3]=='1' else '')).rjust(20) for s in shapes))
for v,d in rows.items():
    print(v.replace('lab_','').ljust(24)+" ".join(("%d/%d %s/%s %s"%(d[s][0][1],d[s][-1][1],d[s][-1][2],d[s][-1][3],d[s][-1][4])).rjust(20) for s in shapes if s in d))
