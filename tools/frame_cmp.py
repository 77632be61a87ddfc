import numpy as np, sys
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
print("differing pixels:", int(d.sum()), "of", d.size)
idx = np.argwhere(d)
for i in idx[:25]:
    i = tuple(i)
    print(i, "block-in-tile", (i[-2] % 16) // 8, (i[-1] % 16) // 8, a[i], b[i], b[i] - a[i])
