import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
ch = bnr_amd.Chain(X, y, 7, 8, 20240501, 1)
ch.init_prior(); ch.run(2, 8, 4)
print("standalone us", ch.debug_time_gram(50))
d = ch.debug_read(430)[400:424].astype(np.int64).reshape(3, 8)
print("realtime start of blocks 100 and 251 relative to block 0 (us):", (d[1][4]-d[0][4])/100.0, (d[2][4]-d[0][4])/100.0, " end:", (d[1][7]-d[0][4])/100.0, (d[2][7]-d[0][4])/100.0, "block0 end", (d[0][7]-d[0][4])/100.0)
for i, blk in enumerate((0, 233, 251)):
    t = d[i]
    print("block", blk, "prologue", t[1]-t[0], "loop", t[2]-t[1], "epilogue", t[3]-t[2], "total", t[3]-t[0], "| realtime ticks (100 MHz) total", t[7]-t[4], "-> WG lifetime %.1f us, shader clock %.2f GHz" % ((t[7]-t[4]) / 100.0, (t[3]-t[0]) / ((t[7]-t[4]) * 10.0)))

a = ch.debug_read(1024).astype(np.int64)
st, en = a[512:512+252], a[768:768+252]
t0 = st.min()
print("WG start times (us) sorted:", np.round(np.sort((st - t0) / 100.0), 1)[::12])
print("WG end times (us) sorted:", np.round(np.sort((en - t0) / 100.0), 1)[::12])
late = np.where((st - t0) / 100.0 > 5)[0]
print("late starters:", len(late), late[:40])
ch.debug_time_gram(1)
a = ch.debug_read(1024).astype(np.int64)
st, en = a[512:512+252], a[768:768+252]
print("stamp kernel before -> first WG start: %.1f us; first WG start -> last WG end: %.1f us; last WG end -> stamp kernel after: %.1f us" % ((st.min() - a[1000]) / 100.0, (en.max() - st.min()) / 100.0, (a[1001] - en.max()) / 100.0))
xcc = (a[256:256+252] >> 32) & 0xF
hw = a[256:256+252] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
for x in range(8):
    m = xcc == x
    if m.sum() == 0: continue
    s0 = st[m].min()
    print("XCC", x, "blocks", int(m.sum()), "max start %.1f" % ((st[m].max() - s0) / 100.0), "ends min/max %.1f %.1f" % ((en[m].min() - s0) / 100.0, (en[m].max() - s0) / 100.0), "distinct (se,sh,cu):", len(set(zip(se[m], sh[m], cu[m]))))
print("single launches (us):", [round(ch.debug_time_gram(1), 1) for _ in range(6)])
print("2 back-to-back (us each):", [round(ch.debug_time_gram(2), 1) for _ in range(3)])
print("10 back-to-back (us each):", round(ch.debug_time_gram(10), 1))
life = (en - st) / 100.0
print("lifetime histogram:", np.histogram(life, bins=[0, 20, 28, 30, 32, 40, 50, 60, 100])[0])
slow = np.where(life > 32)[0]
print("slow blocks:", slow, "xcc", xcc[slow], "life", life[slow])
print("xcc raw values:", sorted(set(((a[256:256+252] >> 32) & 0xFFFF).tolist())))
gm = np.array([0])
for rep in range(6):
    ch.debug_time_gram(1)
    a = ch.debug_read(1024).astype(np.int64)
    st, en = a[512:512+252], a[768:768+252]
    life = (en - st) / 100.0
    hw = a[256:256+252] & 0xFFFFFFFF
    xcc = (a[256:256+252] >> 32) & 0xF
    slow = np.where(life > 32)[0]
    print("rep", rep, "slow blocks", slow.tolist(), "life", np.round(life[slow], 1).tolist(), "xcc", xcc[slow].tolist(), "hwid", [hex(int(h)) for h in hw[slow]], "median life %.1f" % np.median(life))
