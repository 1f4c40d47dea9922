import os, sys, subprocess, tempfile, numpy as np, hashlib
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import test_dtc as T
import conftest
g = np.load(os.path.join(ROOT,"tests","golden","gp_dtc.npz"))
name=sys.argv[3] if len(sys.argv) > 3 else "fa"
X, y, Xu, beta, Xs = T.problem(g, name)
td=tempfile.mkdtemp()
for nm, A in (("X", X), ("y", y), ("Xs", Xs), ("Xu", Xu)):
    T._write_txt(os.path.join(td, nm + ".txt"), A)
exe = os.path.join(ROOT, "gpc_amd", "host", "gp_hosttest")
outs={}
for i in range(int(sys.argv[1])):
    r = subprocess.run([exe, "dtc", td+"/X.txt", td+"/y.txt", td+"/Xs.txt", T._spec(T.CASES[name]), td+"/Xu.txt", "%.17g" % beta, "15"] + ([T.APPROX[name[0]][1]] if name[0] in T.APPROX else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL, timeout=600)
    h=hashlib.md5(b"\n".join(l for l in r.stdout.split(b"\n") if not l.startswith(b"time_"))).hexdigest() + (":rc%d" % r.returncode)
    outs.setdefault(h, []).append(i)
    if len(outs)>1 and len(outs[h])==1:
        v=T._parse(r.stdout.decode()); print("variant", h, "ll_final", v["ll_final"], "ll", v["ll"])
print({h: len(v) for h, v in outs.items()})
# which printed quantities differ between two runs, and by how much
outs2 = []
for i in range(2):
    r = subprocess.run([exe, "dtc", td+"/X.txt", td+"/y.txt", td+"/Xs.txt", T._spec(T.CASES[name]), td+"/Xu.txt", "%.17g" % beta, sys.argv[2] if len(sys.argv) > 2 else "15"] + ([T.APPROX[name[0]][1]] if name[0] in T.APPROX else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL, timeout=600)
    outs2.append(T._parse(r.stdout.decode()))
for k in outs2[0]:
    a, b = np.asarray(outs2[0][k], dtype=float), np.asarray(outs2[1][k], dtype=float)
    if a.shape == b.shape:
        dmax = np.abs(a - b).max() if a.size else 0.0
        print("%-14s size %5d  max |diff| %.3e  (max |value| %.3e)" % (k, a.size, dmax, np.abs(a).max() if a.size else 0.0))
