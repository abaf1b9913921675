import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
print("loss", a["loss"], b["loss"])
rows = []
for k in a.files:
    if k == "loss": continue
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    den = np.abs(x).max() + 1e-30
    rows.append((np.abs(x - y).max() / den, k, x.shape, float(np.linalg.norm(x)), float(np.linalg.norm(y))))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.3e  %-44s %-16s |a|=%.4e |b|=%.4e" % r)
