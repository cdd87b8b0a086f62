import re, sys
txt = sys.stdin.read().splitlines()
head = [l for l in txt if 'launches in' in l]
rows = [l for l in txt if 't=+' in l]
sc, pw = [], []
for l in rows[2:-1]:
    m = re.search(r'\((\d+)Mhz\),1,\(\d+Mhz\),S,([\d.]+)', l)
    if m:
        sc.append(int(m.group(1)))
        pw.append(float(m.group(2)))
print(head[0] if head else "\n".join(txt[-5:]))
if sc:
    print("   sclk %.0f MHz (min %d max %d), package power %.0f W (min %.0f max %.0f), %d samples" % (sum(sc) / len(sc), min(sc), max(sc), sum(pw) / len(pw), min(pw), max(pw), len(sc)))
