"""Multi-GPU exact-GP factorisation: 1-D block-cyclic column panels, one process per GPU.

The reference has no distributed path (its CGp::updateK factors one N x N matrix on one host, CGp.cpp:698-712); this
module is how the same updateK / logLikelihood / posteriorMeanVar arithmetic is spread over the GPUs of one node.

Layout.  The N x N Gram matrix is cut into T = ceil(N / nb) column panels of nb columns.  Panel j lives on rank
j % P, all of its rows on that rank, so every rank holds ~N^2 / P doubles in ONE column-major array whose local panel
l is global panel rank + l * P.  Nothing of K ever exists in one place: every rank generates its panels directly
from X (gpc_gram_block_f64; X is N x D and replicated).  Below the N matrix rows the array carries E "extra rows":
right-hand sides stored transposed (y' for the likelihood, K(X*, X) for the predictive variance).  They ride through
the factorisation like any other row, which turns them into (L^-1 y)' and (L^-1 K(X, X*))' for free -- the forward
substitutions of CGp::updateAlpha (CGp.cpp:469-489) and CGp::posteriorMeanVar (CGp.cpp:585-607) need no extra pass.

Right-looking step k (owner = k % P):
    owner   : gpc_potrf_panel_f64 on its panel (diagonal dpotrf + dtrsm of everything below, extra rows included),
              pack it into a contiguous buffer, broadcast it (RCCL over xGMI; the only data-path collective);
    everyone: gpc_syrk_blockcyclic_f64 -- ONE launch updates all local panels right of k with the received panel.
Look-ahead: the rank that owns panel k+1 updates that panel first (U1), then factors and broadcasts it on a second,
high-priority stream while the rest of the update (U2) keeps all CUs busy; receive buffers are double-buffered.
Communication volume is N^2/2 doubles per rank in total, in T large messages.

torch is plumbing here (device memory, streams, torch.distributed); all arithmetic goes through `ops`, by default
the C-ABI of libgpc_hip.so (HipOps).  The CPU test-suite drives the same orchestration over gloo with a numpy stand-in
for `ops` that lives under tests/ -- this module never falls back to it.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

LOG2PI = math.log(2.0 * math.pi)


class HipOps(object):
    """Local arithmetic = libgpc_hip.so on torch's current HIP stream.  No CPU path."""

    def __init__(self, device=None):
        from . import api
        if not torch.cuda.is_available():
            raise RuntimeError("gpc_amd.dist.HipOps needs a gfx950 GPU: there is no CPU fallback")
        api.lib()
        self.api = api
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def empty(self, rows, cols):
        return self.api.empty(rows, cols, self.device)

    def zeros(self, rows, cols):
        return self.api.zeros(rows, cols, self.device)

    def from_host(self, a):
        return self.api.from_host(a, self.device)

    def kspec(self, terms):
        return self.api.kspec(terms)

    def info_word(self):
        return torch.zeros(1, dtype=torch.int32, device=self.device)

    def gram_block(self, ks, X, i0, m, j0, n, out):
        self.api.gram_block(ks, X, i0, m, j0, n, out=out)

    def gram_cross(self, ks, X, X2, out):
        self.api.gram_cross(ks, X, X2, out=out)

    def gram_diag(self, ks, X):
        return self.api.gram_diag(ks, X)

    def potrf_panel(self, panel, col0, info):
        self.api.potrf_panel(panel, col0, info)

    def syrk_blockcyclic(self, P, C, row0, j0, pstride, nb):
        self.api.syrk_blockcyclic(P, C, row0, j0, pstride, nb)

    def logdet_chol(self, Ljj):
        return self.api.logdet_chol(Ljj)

    def colnorm2(self, A):
        return self.api.colnorm2(A)

    def gemm(self, A, B, C, transa, transb, alpha, beta):
        self.api.gemm(A, B, C, transa=transa, transb=transb, alpha=alpha, beta=beta)

    def trsm(self, A, B, trans):
        self.api.trsm(A, B, side="L", uplo="L", trans=trans, diag="N")

    def trace(self, A):
        return self.api.trace(A)

    def add_diag(self, A, c):
        self.api.add_diag_(A, c)


class _NoStream(object):
    """Stream/event stand-in for CPU tensors (gloo tests): everything is synchronous."""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def wait_event(self, ev):
        pass

    def record_event(self, ev=None):
        return ev


class _Streams(object):
    def __init__(self, device, single_stream=False):
        self.cuda = device.type == "cuda"
        if self.cuda:
            self.main = torch.cuda.current_stream(device)
            if single_stream:          # safe mode: panel work and collectives are serialised on the main stream
                self.panel = self.main
            else:
                self.panel = torch.cuda.Stream(device=device, priority=-1)
                self.panel.wait_stream(self.main)
        else:
            self.main = self.panel = _NoStream()

    def on_panel(self):
        return torch.cuda.stream(self.panel) if self.cuda else self.panel

    def event(self):
        return torch.cuda.Event() if self.cuda else None

    def record(self, stream):
        if not self.cuda:
            return None
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def wait(self, stream, ev):
        if self.cuda and ev is not None:
            stream.wait_event(ev)


def default_nb():
    return int(os.environ.get("GPC_DIST_NB", "512"))


class _Works(object):
    """Several asynchronous collectives waited for as one."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


class DistGp(object):
    """Block-cyclic counterpart of CGp's FTC state: factor of K (distributed), log|K|, L^-1 y, and what follows.

    terms: kernel as for api.kspec; X (N x D) and y (N x d) are host numpy arrays, identical on every rank.
    Xstar (optional, Ns x D): test inputs whose predictive variance is wanted (carried as extra rows).
    """

    def __init__(self, terms, X, y=None, Xstar=None, nb=None, ops=None, group=None, sync=None):
        # sync (default: env GPC_DIST_SYNC): no second stream, blocking collectives -- no look-ahead, but nothing about
        # the ordering is left to events.  bench.py falls back to it if its start-up self-check of the overlapped
        # mode disagrees with a single-GPU factorisation.
        self.sync = (os.environ.get("GPC_DIST_SYNC", "0") == "1") if sync is None else bool(sync)
        self.ops = ops if ops is not None else HipOps()
        self.group = group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.nb = int(nb) if nb else default_nb()
        # collectives are skipped in a 1-rank job unless GPC_DIST_FORCE_COMM=1 (exercises the RCCL calls on one GPU)
        self.comm = self.P > 1 or (dist.is_initialized() and os.environ.get("GPC_DIST_FORCE_COMM", "0") == "1")
        # width of the column slabs a panel is factored and broadcast in (0 = whole panel in one piece)
        self.slab = int(os.environ.get("GPC_DIST_SLAB", "128"))
        if self.slab % 128 != 0 or self.slab < 0:
            raise ValueError("GPC_DIST_SLAB must be 0 or a multiple of 128")
        if self.nb % 128 != 0:
            raise ValueError("panel width must be a multiple of 128")
        X = np.asarray(X, dtype=np.float64)
        self.N, self.D = X.shape
        self.T = (self.N + self.nb - 1) // self.nb
        self.terms = terms
        self.ks = self.ops.kspec(terms)
        self.X = self.ops.from_host(X)
        self.y = None if y is None else self.ops.from_host(np.asarray(y, dtype=np.float64).reshape(self.N, -1))
        self.d = 0 if self.y is None else self.y.shape[1]
        self.Xs = None if Xstar is None else self.ops.from_host(np.asarray(Xstar, dtype=np.float64))
        self.Ns = 0 if self.Xs is None else self.Xs.shape[0]
        self.E = self.d + self.Ns
        self.Mtot = self.N + self.E + ((self.N + self.E) & 1)     # even row count (16-byte aligned columns); pad row = 0
        self.mine = list(range(self.rank, self.T, self.P))            # global indices of the local panels
        self.ncols = sum(self.width(j) for j in self.mine)
        self.A = self.ops.empty(self.Mtot, max(self.ncols, 1))
        self.A[self.N:, :].zero_()
        self.st = _Streams(self.A.device, single_stream=self.sync)
        # two receive buffers (panel k uses k % 2); flat so that every step's M x w panel is one contiguous message
        self.buf = [torch.empty(self.Mtot * self.nb, dtype=torch.float64, device=self.A.device) for _ in range(2)]
        self.info = self.ops.info_word()
        self.logdet = None
        self.factored = False

    # ---- index helpers ----------------------------------------------------------------------------------------------
    def width(self, j):
        return min(self.nb, self.N - j * self.nb)

    def owner(self, j):
        return j % self.P

    def lcol(self, j):
        """first local column of (owned) global panel j"""
        return (j // self.P) * self.nb

    def first_local_after(self, k):
        """local index of the first local panel whose global index exceeds k"""
        return 0 if k < self.rank else (k - self.rank) // self.P + 1

    def panel_view(self, j):
        """rows j*nb.. (matrix + extra rows) of owned panel j: the tall panel with its diagonal block on top"""
        c = self.lcol(j)
        return self.A[j * self.nb:, c:c + self.width(j)]

    def buf_view(self, k):
        M, w = self.Mtot - k * self.nb, self.width(k)
        flat = self.buf[k % 2][:M * w]
        return flat, flat.view(w, M).t()

    # ---- Gram generation ----------------------------------------------------------------------------------------------
    def fill(self):
        """Every rank builds its own panels from X: lower part of K, then y' and K(X*, X) underneath."""
        N, nb = self.N, self.nb
        for j in self.mine:
            w, c = self.width(j), self.lcol(j)
            self.ops.gram_block(self.ks, self.X, j * nb, N - j * nb, j * nb, w, self.A[j * nb:N, c:c + w])
            if self.d:
                self.A[N:N + self.d, c:c + w].copy_(self.y[j * nb:j * nb + w, :].t())
            if self.Ns:
                self.ops.gram_cross(self.ks, self.Xs, self.X[j * nb:j * nb + w, :],
                                    self.A[N + self.d:N + self.E, c:c + w])
        self.factored = False

    # ---- communication ------------------------------------------------------------------------------------------------
    def _bcast(self, flat, src):
        """Broadcast `flat` from rank src.  RCCL: asynchronous on its own stream, ordered after the current stream;
        gloo (CPU tests, or GPU tensors staged through the host): synchronous.  Returns a work handle or None."""
        if not self.comm:
            return None
        gsrc = src if self.group is None else dist.get_global_rank(self.group, src)
        if self.backend == "nccl":
            if self.sync:
                dist.broadcast(flat, gsrc, group=self.group)       # the current stream waits for it
                return None
            return dist.broadcast(flat, gsrc, group=self.group, async_op=True)
        if flat.is_cuda:
            h = flat.cpu() if self.rank == src else torch.empty(flat.shape, dtype=flat.dtype)
            dist.broadcast(h, gsrc, group=self.group)
            if self.rank != src:
                flat.copy_(h)
        else:
            dist.broadcast(flat, gsrc, group=self.group)
        return None

    def _allreduce_sum(self, t):
        if not self.comm:
            return t
        if self.backend == "nccl" or not t.is_cuda:
            dist.all_reduce(t, group=self.group)
            return t
        h = t.cpu()
        dist.all_reduce(h, group=self.group)
        t.copy_(h)
        return t

    # ---- factorisation ------------------------------------------------------------------------------------------------
    def _factor_and_send(self, k, after):
        """On the panel stream: (owner) wait for `after`, factor panel k, pack; (all) broadcast.  Returns what the
        main stream has to wait for before it may read the panel."""
        st = self.st
        with st.on_panel():
            prev = self._inflight[k % 2]
            if prev is not None:                      # the buffer's previous message must have left / been consumed
                if prev[0] is not None:
                    prev[0].wait()
                st.wait(st.panel, self._free[k % 2])
            flat, view = self.buf_view(k)
            ev_fact = None
            own = self.owner(k) == self.rank
            w, M = self.width(k), self.Mtot - k * self.nb
            sb = self.slab if (self.comm and self.slab > 0) else w
            works = []
            after_first, after_rest = after if isinstance(after, tuple) else (after, None)
            if own:
                st.wait(st.panel, after_first)
                pv = self.panel_view(k)
            # The panel leaves in column slabs: slab s is final as soon as it is factored, so its broadcast overlaps
            # the factorisation of slabs s+1.. (a slab of whole columns is one contiguous piece of the buffer).
            for c0 in range(0, w, sb):
                ws = min(sb, w - c0)
                if own:
                    self.ops.potrf_panel(pv[c0:, c0:c0 + ws], k * self.nb + c0, self.info)
                    if self.comm:
                        view[:, c0:c0 + ws].copy_(pv[:, c0:c0 + ws])
                wk = self._bcast(flat[c0 * M:(c0 + ws) * M], self.owner(k))
                if wk is not None:
                    works.append(wk)
                if own and c0 + ws < w:
                    # right-looking step inside the panel: the columns still to be factored take this slab's update
                    if after_rest is not None:
                        st.wait(st.panel, after_rest)     # ... once the previous panel's update has reached them
                        after_rest = None
                    r0 = k * self.nb + c0 + ws
                    self.ops.syrk_blockcyclic(pv[c0 + ws:, c0:c0 + ws], pv[c0 + ws:, c0 + ws:w], r0, r0 // sb, 1, sb)
            if own:
                ev_fact = st.record(st.panel)
            work = _Works(works) if works else None
            ev_recv = st.record(st.panel)
        self._inflight[k % 2] = (work, ev_recv)
        return work, ev_recv, ev_fact

    def factor(self):
        """Right-looking block-cyclic Cholesky with depth-1 look-ahead.  Returns LAPACK-style info (0 = ok)."""
        st, nb, T, P, r = self.st, self.nb, self.T, self.P, self.rank
        self._inflight = [None, None]
        self._free = [None, None]
        self.info.zero_()
        ready = st.record(st.main)                    # Gram generation precedes everything on the panel stream
        pending = self._factor_and_send(0, ready)
        for k in range(T):
            work, ev_recv, ev_fact = pending
            own = self.owner(k) == r
            if own:                                   # the owner reads its own storage: no need to wait for the wire
                st.wait(st.main, ev_fact)
                src = self.panel_view(k)
            else:
                if work is not None:
                    work.wait()
                st.wait(st.main, ev_recv)
                src = self.buf_view(k)[1]
            row0 = (k + 1) * nb
            l0 = self.first_local_after(k)
            c0 = l0 * nb
            Pk = src[nb:, :] if row0 < self.Mtot else None
            u1 = None
            if k + 1 < T and self.owner(k + 1) == r:
                # U1: the next panel first, so that its factorisation overlaps the rest of this update
                w1 = self.width(k + 1)
                sa = min(self.slab, w1) if (self.comm and self.slab > 0) else w1
                # ... and its first slab before the rest of it: that slab's factorisation is what the chain waits for
                self.ops.syrk_blockcyclic(Pk, self.A[row0:, c0:c0 + sa], row0, k + 1, P, nb)
                u1 = st.record(st.main)
                if sa < w1:
                    self.ops.syrk_blockcyclic(Pk, self.A[row0:, c0 + sa:c0 + w1], row0, (row0 + sa) // self.slab, 1,
                                              self.slab)
                    u1 = (u1, st.record(st.main))
                c0 += w1
                l0 += 1
            if k + 1 < T:
                pending = self._factor_and_send(k + 1, u1)
            if c0 < self.ncols:
                self.ops.syrk_blockcyclic(Pk, self.A[row0:, c0:self.ncols], row0, r + l0 * P, P, nb)
            self._free[k % 2] = st.record(st.main)
        if st.cuda:
            st.main.wait_stream(st.panel)
        info = self.info.to(torch.int64)
        if self.comm:                                 # smallest positive info over the ranks
            big = torch.where(info > 0, info, torch.full_like(info, 1 << 60))
            if self.backend == "nccl" or not big.is_cuda:
                dist.all_reduce(big, op=dist.ReduceOp.MIN, group=self.group)
            else:
                h = big.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MIN, group=self.group)
                big = h
            info = torch.where(big < (1 << 60), big, torch.zeros_like(big))
        self.factored = int(info.item()) == 0
        return int(info.item())

    def _add_diag(self, c):
        """K(i,i) += c on the locally owned diagonal blocks (CMatrix::addDiag, the jitter of jitChol)."""
        for j in self.mine:
            w = self.width(j)
            self.ops.add_diag(self.panel_view(j)[:w, :], c)

    def _trace(self):
        """trace(K) from the locally owned diagonal blocks (all ranks get the global sum)."""
        s = 0.0
        for j in self.mine:
            w = self.width(j)
            s += self.ops.trace(self.panel_view(j)[:w, :])
        t = torch.tensor([s], dtype=torch.float64, device=self.A.device)
        return float(self._allreduce_sum(t).item())

    def update_k(self, max_tries=20):
        """CGp::updateK (FTC) on the distributed matrix: Gram, factor, log|K|, with CMatrix::jitChol's jitter schedule
        (CMatrix.cpp:767-804) when a pivot fails: first candidate 1e-6 * trace(K) / N, x10 per retry, accumulated on the
        diagonal of a regenerated K; gives up when the candidate exceeds 10 or after 20 tries."""
        self.fill()
        jitter, total, tries = None, 0.0, 0
        while True:
            if jitter is None:
                trace = self._trace()              # one scalar all-reduce; K is still intact here
                jitter = 1e-6 * trace / float(self.N)
            info = self.factor()
            if info == 0:
                break
            total += jitter
            jitter *= 10.0
            tries += 1
            if jitter > 10.0 or tries >= max_tries:
                raise ArithmeticError("distributed Cholesky: leading minor %d is not positive definite after %d jitter "
                                      "steps (total %g)" % (info, tries, total))
            self.fill()
            self._add_diag(total)
        self.jitter = total
        s = 0.0
        for j in self.mine:
            w = self.width(j)
            s += self.ops.logdet_chol(self.panel_view(j)[:w, :])
        t = torch.tensor([s], dtype=torch.float64, device=self.A.device)
        self.logdet = float(self._allreduce_sum(t).item())
        return self.logdet

    # ---- what CGp reads off the factor ----------------------------------------------------------------------------------
    def _extra_rowsumsq(self, e0, e1):
        """sum over ALL columns (all ranks) of the squares of extra rows e0..e1 -> (e1-e0) device vector"""
        ne = e1 - e0
        if self.ncols:
            blk = self.ops.empty(self.ncols, ne)
            blk.copy_(self.A[self.N + e0:self.N + e1, :self.ncols].t())       # transposed copy: rows become columns
            s = self.ops.colnorm2(blk)
        else:
            s = self.ops.zeros(ne, 1)
        return self._allreduce_sum(s.reshape(-1).contiguous())

    def log_likelihood(self):
        """CGp::logLikelihood, FTC branch (CGp.cpp:913-938, 1002-1013): -0.5 (sum_j |L^-1 y_j|^2 + d log|K|) - d N/2 log 2 pi."""
        assert self.factored and self.d > 0
        quad = float(self._extra_rowsumsq(0, self.d).sum().item())
        return -0.5 * (quad + self.d * self.logdet) - self.d * self.N * 0.5 * LOG2PI

    def alpha(self):
        """K^-1 y (N x d, replicated on every rank): column-oriented back substitution L' alpha = L^-1 y over the
        panels, last to first; the owner of panel k holds every L(j,k), j > k, it needs (CGp::updateAlpha)."""
        assert self.factored and self.d > 0
        N, nb, d = self.N, self.nb, self.d
        al = self.ops.zeros(N, d)
        for k in range(self.T - 1, -1, -1):
            w = self.width(k)
            t = self.ops.empty(w, d)
            if self.owner(k) == self.rank:
                pv = self.panel_view(k)
                t.copy_(pv[N - k * nb:N - k * nb + d, :].t())                 # z_k = (L^-1 y)(panel k rows)
                if (k + 1) * nb < N:
                    self.ops.gemm(pv[w:N - k * nb, :], al[(k + 1) * nb:, :], t, "T", "N", -1.0, 1.0)
                self.ops.trsm(pv[:w, :], t, "T")
            if self.comm:
                flat = t.t().contiguous().reshape(-1)
                if self.owner(k) != self.rank:
                    flat = torch.empty(w * d, dtype=torch.float64, device=self.A.device)
                work = self._bcast(flat, self.owner(k))
                if work is not None:
                    work.wait()
                t = flat.view(d, w).t()
            al[k * nb:k * nb + w, :].copy_(t)
        return al

    def posterior(self, alpha=None):
        """CGp::posteriorMeanVar before output scale/bias (CGp.cpp:548-625, 642-663): mu = K(X*,X) alpha,
        var = k(x*,x*) - |L^-1 K(X,x*)|^2.  Returns device tensors (Ns x d, Ns)."""
        assert self.factored and self.Ns > 0
        if alpha is None:
            alpha = self.alpha()
        kx = self.ops.empty(self.Ns, self.N)
        self.ops.gram_cross(self.ks, self.Xs, self.X, kx)
        mu = self.ops.zeros(self.Ns, self.d)
        self.ops.gemm(kx, alpha, mu, "N", "N", 1.0, 0.0)
        var = self.ops.gram_diag(self.ks, self.Xs).reshape(-1) - self._extra_rowsumsq(self.d, self.E)
        return mu, var

    def gather_factor(self):
        """Debug/tests only: assemble the full lower factor on every rank's HOST (N x N numpy)."""
        L = np.zeros((self.N, self.N))
        for j in range(self.T):
            w = self.width(j)
            blk = torch.zeros((w, self.N - j * self.nb), dtype=torch.float64)
            if self.owner(j) == self.rank:
                blk.copy_(self.panel_view(j)[:self.N - j * self.nb, :].t())
            if self.P > 1:
                dist.broadcast(blk, self.owner(j) if self.group is None else
                               dist.get_global_rank(self.group, self.owner(j)), group=self.group)
            L[j * self.nb:, j * self.nb:j * self.nb + w] = blk.numpy().T
        return np.tril(L)
