# How long do open() and stat() take on this box's file system?  (The compress pipeline's start-up is dominated by it.)
import os, sys, time, tempfile, threading
d = tempfile.mkdtemp(prefix="zwz_open_", dir=os.environ.get("TMPDIR", "/tmp"))
N = 8000
for i in range(N):
    with open(os.path.join(d, "f%05d.bin" % i), "wb") as f: f.write(b"x" * 1000)
names = [os.path.join(d, "f%05d.bin" % i) for i in range(N)]
t = time.perf_counter(); [os.stat(n) for n in names]; t_stat = time.perf_counter() - t
t = time.perf_counter(); fds = [os.open(n, os.O_RDONLY) for n in names]; t_open = time.perf_counter() - t
t = time.perf_counter(); [os.close(f) for f in fds]; t_close = time.perf_counter() - t
def worker(lo, hi, out):
    out.extend(os.open(n, os.O_RDONLY) for n in names[lo:hi])
t = time.perf_counter(); outs = [[] for _ in range(8)]
th = [threading.Thread(target=worker, args=(i * N // 8, (i + 1) * N // 8, outs[i])) for i in range(8)]
[x.start() for x in th]; [x.join() for x in th]; t_open8 = time.perf_counter() - t
for o in outs: [os.close(f) for f in o]
print("per file: stat %.1f us, open %.1f us (8 threads: %.1f us), close %.1f us" % (t_stat / N * 1e6, t_open / N * 1e6, t_open8 / N * 1e6, t_close / N * 1e6))
for n in names: os.unlink(n)
os.rmdir(d)
