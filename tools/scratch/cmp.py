import numpy as np, sys
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
ys, xs = np.nonzero(d)
print("differing pixels:", d.sum())
for y, x in list(zip(ys, xs))[:40]:
    print(y, x, "block", y // 8, x // 8, "in-block", y % 8, x % 8, a[y, x], b[y, x], (b[y, x] - a[y, x]))
