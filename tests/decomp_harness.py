"""TEST HARNESS (moved out of the product package in round 3): the x-slab decomposition restated in Python over torch.distributed.

The product runs the decomposition inside the C library (csrc/fdw_slabs.cpp + csrc/fdw_comm.cpp: RCCL, ranks-as-threads, or the
process transport); this file keeps the same scheme with a pluggable stepper so that the exchange logic can be exercised WITHOUT a GPU --
gloo, world_size 2/3, the CPU oracle as the compute kernel (tests/test_decomp_cpu.py) -- and as an independent second implementation the
C driver is compared with on one GPU (tests/test_gpu_parity.py).  Nothing under parallel_finite_difference_computation_amd/ or bench.py
imports it.

x-slab domain decomposition of the forward time loop across the GPUs of one node.

The reference has no multi-GPU path; its only parallel axis besides shots is
space.  Rows (x, the slow axis) are split into `world` contiguous slabs, one process per GPU.  A
ghost row is `pitch` contiguous floats, so a halo is one contiguous block.

Deep halos: with half order h and `ksteps` steps per exchange every interior side carries
G = h*ksteps ghost rows.  Right after an exchange all local rows are valid; step j of the cycle
(j = 1..ksteps) can only update rows [h*j, nxl - h*j) on the interior sides, so the valid region
shrinks by h per step and is exactly the owned rows after ksteps steps.  The redundant work is
h*(ksteps-1)/2 rows per side per step; in exchange the per-step collective (a 2x128 KiB message on a
~25 us step, latency bound over xGMI) becomes one 2*G-row message every ksteps steps.

Both time levels travel (leap-frog state is the pair (p, pp)); what is exchanged is the raw memory
state of the owner's rows, so the "lazy taper" bookkeeping of the kernels (csrc/fdw_device.h) stays a
pure function of (memory, step index) on every rank.  Arithmetic per point is unchanged, hence the
decomposed result is bit-identical to the single-slab result.

The stepper is pluggable so that the exchange logic can be exercised on CPU (gloo, world_size 2)
with the oracle as the compute kernel (tests only); the product path uses `HipSlabStepper`.

`SlabBack` decomposes the backward loop the same way (fd-code.cu:302-339): FOUR fields travel per exchange (the source-field pair
that is reconstructed backwards in time and the receiver-field pair), receiver injection and the imaging condition are local to the
rows a slab holds, the image never travels and only its owned rows are meaningful.
"""
import torch
import torch.distributed as dist

from parallel_finite_difference_computation_amd.decomp import SlabGeometry, slab_bounds  # noqa: F401


class HipSlabStepper:
    """Product stepper: one FDWave slab context, fused forward step on the given rows."""

    def __init__(self, fdwave_ctx):
        self.ctx = fdwave_ctx

    def step(self, d_p, d_pp, d_v2, r0, r1, it, first, d_srce, sx, sz, stream):
        from parallel_finite_difference_computation_amd._lib import MODE_FWD
        inj = d_srce.data_ptr() + 4 * it if d_srce is not None else None
        self.ctx.dev_step(MODE_FWD, d_p.data_ptr(), d_pp.data_ptr(), d_v2.data_ptr(), r0, r1, pp_twice=not first,
                          d_inj=inj, inj_x=sx if d_srce is not None else -1, inj_z=sz, stream=stream)


    def steps_shrink(self, d_p, d_pp, d_v2, it0, nsteps, first, d_srce, sx, sz, j0, shrink_lo, shrink_hi, stream):
        """nsteps consecutive cycle steps j0.. in ONE library call (role swaps and shrinking ranges in C)."""
        self.ctx.dev_steps_shrink(d_p.data_ptr(), d_pp.data_ptr(), d_v2.data_ptr(), d_srce.data_ptr() if d_srce is not None else None,
                                  sx, sz, it0, nsteps, not first, j0, shrink_lo, shrink_hi, stream=stream)


class _SlabLoop:
    """What the forward and the backward slab drivers share: streams, the halo exchange of `exchange_fields()` (by ROLE, so that the
    message order is the same on every rank) and the cycle loop.  A subclass provides exchange_fields() and cycle(kk, more_after, stream)."""

    def _init_loop(self, geom, group, overlap, cuda, side_stream):
        self.g, self.group, self.cuda, self.overlap = geom, group, cuda, overlap
        self._ops = {}
        self.fresh = False          # ghosts of all travelling fields are up to date
        if geom.world > 1 and (geom.o1 - geom.o0) < 2 * geom.G:
            self.overlap = False    # strips would collide: fall back to exchange-then-compute
        self._send_after = None     # stream whose queued work the next exchange has to wait for (default: compute)
        self.side = None
        if self.cuda:
            self.compute = torch.cuda.Stream()
            self.comm = torch.cuda.Stream()
            self.side = torch.cuda.Stream() if side_stream else None   # boundary strips of a split pass
            torch.cuda.synchronize()    # whatever filled the fields (another stream) must have landed before these streams touch them

    # ---- halo exchange ------------------------------------------------------------------------
    def _exchange_ops(self):
        """P2P descriptors for every travelling field, built once per buffer (the views alias fixed memory)."""
        g, out = self.g, []
        for f in self.exchange_fields():         # by ROLE: every rank is in the same state, so the message order matches
            key = f.data_ptr()
            if key not in self._ops:
                ops = []
                if g.has_lo:
                    s0, s1 = g.send_lo()
                    r0, r1 = g.recv_lo()
                    ops.append(dist.P2POp(dist.isend, f[s0:s1], g.rank - 1, group=self.group))
                    ops.append(dist.P2POp(dist.irecv, f[r0:r1], g.rank - 1, group=self.group))
                if g.has_hi:
                    s0, s1 = g.send_hi()
                    r0, r1 = g.recv_hi()
                    ops.append(dist.P2POp(dist.isend, f[s0:s1], g.rank + 1, group=self.group))
                    ops.append(dist.P2POp(dist.irecv, f[r0:r1], g.rank + 1, group=self.group))
                self._ops[key] = ops
            out += self._ops[key]
        return out

    def exchange(self, wait_compute=True):
        """Refresh the ghost rows of the travelling fields.  On GPU the transfer runs on the comm stream: it
        starts after everything already queued on the compute stream (or on the stream recorded in
        `self._send_after`) and the compute stream is NOT made to wait here."""
        ops = self._exchange_ops()
        if not ops:
            return
        if self.cuda:
            if wait_compute:
                self.comm.wait_stream(self._send_after if self._send_after is not None else self.compute)
                self._send_after = None
            stream_aware = dist.get_backend(self.group) == "nccl"
            if not stream_aware:
                # rehearsal backends (gloo stages CUDA tensors through host copies on threads/streams of its own):
                # fence the whole device on both sides so that only the exchange LOGIC is exercised
                torch.cuda.synchronize()
            with torch.cuda.stream(self.comm):
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if not stream_aware:
                torch.cuda.synchronize()
        else:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self.fresh = True

    def run(self, nsteps):
        """nsteps iterations.  Cycle = exchange, then ksteps iterations on shrinking row ranges.
        With overlap the exchange that opens the NEXT cycle is started as soon as the boundary
        strips of the cycle's last iteration exist, and runs beside that iteration's interior rows."""
        g = self.g
        stream = self.compute.cuda_stream if self.cuda else None
        done = 0
        while done < nsteps:
            kk = min(g.ksteps, nsteps - done)
            for tag in self.cycle(kk, done + kk < nsteps, stream):
                if g.world == 1:
                    continue
                if tag == "pre":
                    if not self.fresh:
                        self.exchange()
                    if self.cuda:
                        self.compute.wait_stream(self.comm)    # ghosts must have landed before they are read
                    self.fresh = False
                else:
                    self.exchange()
            done += kk

    def synchronize(self):
        if self.cuda:
            self.compute.synchronize()
            self.comm.synchronize()
            if self.side is not None:
                self.side.synchronize()

    def owned(self, f):
        """The owned rows of a local field (drops ghosts)."""
        return f[self.g.g_lo:self.g.nxl - self.g.g_hi]


class SlabForward(_SlabLoop):
    """fd_forward's loop (fd-code.cu:259-267) on one slab of a decomposed grid."""

    PIPE = 4          # time steps per pass of the wave-pipeline kernel (fdw_dev_step4)

    def __init__(self, geom, stepper, fields, v2, srce=None, sx=-1, sz=0, group=None, overlap=True, pipe_ctx=None):
        """fields: two [nxl][pitch] tensors whose roles swap every step, or FOUR when `pipe_ctx` (the slab's FDWave
        context) is given: full cycles then go four steps per pass through the wave-pipeline kernel, out of place over the
        four rotating buffers (needs ksteps % 4 == 0 and order 8)."""
        self.stepper = stepper
        self.bufs = list(fields)
        self.a, self.b = self.bufs[0], self.bufs[1]
        self.pipe_ctx = pipe_ctx if (pipe_ctx is not None and len(self.bufs) == 4 and geom.ksteps % self.PIPE == 0 and geom.h == 4) else None
        self.v2, self.srce, self.sx, self.sz = v2, srce, sx, sz
        self.it = 0
        self.d_p, self.d_pp = self.a, self.b       # the reference's (d_p, d_pp) BEFORE its swap: d_pp is the newest field
        self._init_loop(geom, group, overlap, self.a.is_cuda, self.pipe_ctx is not None)

    def exchange_fields(self):
        return (self.d_p, self.d_pp)

    # ---- time loop ----------------------------------------------------------------------------
    def _step(self, r0, r1, stream):
        if r1 > r0:
            self.stepper.step(self.d_p, self.d_pp, self.v2, r0, r1, self.it, self.it == 0, self.srce, self.sx, self.sz, stream)

    def cycle(self, kk, more_after, stream):
        """One cycle as a generator: yields "pre" where the ghosts must be valid (cycle start) and "mid"
        where the NEXT cycle's exchange can start (overlap).  `run` drives it with RCCL; tests drive several
        slabs in lockstep on one GPU with in-process copies at the yields."""
        g = self.g
        yield "pre"
        if self.pipe_ctx is not None and kk == g.ksteps:
            yield from self._pipe_cycle(more_after, stream)
            return
        split_last = self.overlap and g.world > 1 and kk == g.ksteps and more_after
        nbulk = kk - 1 if split_last else kk
        j_next = 1
        if nbulk > 0 and hasattr(self.stepper, "steps_shrink"):
            # the plain steps of the cycle in one library call (keeps the host ahead of the GPU)
            self.stepper.steps_shrink(self.d_p, self.d_pp, self.v2, self.it, nbulk, self.it == 0, self.srce, self.sx, self.sz,
                                      1, g.has_lo, g.has_hi, stream)
            if nbulk % 2:
                self.d_p, self.d_pp = self.d_pp, self.d_p
            self.it += nbulk
            j_next = nbulk + 1
        for j in range(j_next, kk + 1):
            self.d_p, self.d_pp = self.d_pp, self.d_p      # fd-code.cu:260-262
            r0, r1 = g.update_range(j)
            if split_last and j == kk:
                # r0 == g_lo and r1 == nxl - g_hi here: the strips are the rows the neighbours need
                lo_end = r0 + g.G if g.has_lo else r0
                hi_beg = r1 - g.G if g.has_hi else r1
                self._step(r0, lo_end, stream)
                self._step(hi_beg, r1, stream)
                yield "mid"                                 # exchange on the comm stream: waits for the strips only
                self._step(lo_end, hi_beg, stream)          # interior, concurrent with the transfer
            else:
                self._step(r0, r1, stream)
            self.it += 1

    def _pipe_cycle(self, more_after, stream):
        """A full cycle, four steps per pass (fdw_dev_step4).  Pass j = 1..ksteps/4 produces the rows still valid on the
        interior sides, [16j, nxl - 16j); the last pass does the two strips the neighbours need first (one launch, short
        chunks), lets the exchange of the next cycle start ("mid") and then does the interior beside the transfer."""
        g, ctx, P = self.g, self.pipe_ctx, self.PIPE
        passes = g.ksteps // P
        split_last = self.overlap and g.world > 1 and more_after and (g.o1 - g.o0) >= 2 * g.G + 16
        v2p = self.v2.data_ptr()
        for j in range(1, passes + 1):
            spare = [b for b in self.bufs if b is not self.d_p and b is not self.d_pp]
            out1, out2 = spare[0], spare[1]
            p_in, pp_in = self.d_pp, self.d_p                    # the kernel's p is the newest field (the reference's d_p after its swap)
            lo = P * g.h * j if g.has_lo else 0
            hi = g.nxl - (P * g.h * j if g.has_hi else 0)
            srce_it = self.srce.data_ptr() + 4 * self.it if self.srce is not None else None
            common = dict(pp_twice=self.it > 0, d_srce_it=srce_it, sx=self.sx if self.srce is not None else -1, sz=self.sz, stream=stream)
            args = (p_in.data_ptr(), pp_in.data_ptr(), v2p, out1.data_ptr(), out2.data_ptr())
            if j == passes and split_last:
                ra = (lo, lo + g.G) if g.has_lo else (0, 0)
                rb = (hi - g.G, hi) if g.has_hi else (0, 0)
                if not g.has_lo:
                    ra, rb = rb, (0, 0)
                # the strips go to a stream of their own: their few workgroups are a latency chain that fits beside the interior launch
                if self.cuda:
                    self.side.wait_stream(self.compute)
                    strips = dict(common, stream=self.side.cuda_stream)
                    self._send_after = self.side
                else:
                    strips = common
                ctx.dev_step4(*args, r0=ra[0], r1=ra[1], r0b=rb[0], r1b=rb[1], xchunk=23, **strips)
                self.d_p, self.d_pp = out1, out2                 # what the exchange started at "mid" sends and fills
                yield "mid"
                ctx.dev_step4(*args, r0=lo + g.G if g.has_lo else lo, r1=hi - g.G if g.has_hi else hi, **common)
            else:
                ctx.dev_step4(*args, r0=lo, r1=hi, **common)
                self.d_p, self.d_pp = out1, out2
            self.it += P

    def run(self, nsteps):
        """nsteps forward iterations; returns the reference's (d_p, d_pp) after the loop."""
        super().run(nsteps)
        return self.d_p, self.d_pp


class HipSlabBackStepper:
    """Product stepper of the backward loop: one iteration of fd_back on the given rows through the slab's FDWave context
    (one launch: source-field step + receiver step + injection + imaging fused)."""

    def __init__(self, fdwave_ctx):
        self.ctx = fdwave_ctx

    def back_iter(self, step_source, f1, f0, pr, ppr, v2, r0, r1, it, samples, gz, img, stream):
        nx = self.ctx.nx
        self.ctx.dev_back_iter(step_source, f1.data_ptr(), f0.data_ptr(), pr.data_ptr(), ppr.data_ptr(), v2.data_ptr(), r0, r1, it > 0,
                               samples.data_ptr() + 4 * nx * it, gz, img.data_ptr(), stream=stream)


class SlabBack(_SlabLoop):
    """fd_back's loop (fd-code.cu:302-339) on one slab of a decomposed grid.

    State per slab: the source-field pair (F_{k-1}, F_{k-2}) -- before iteration 2 these are the forward pass's two snapshots
    (fd-code.cu:304-314) -- the receiver-field pair (r^k, r^{k-1}) and the image accumulator on the slab's rows.  An iteration
    reconstructs F_k (from iteration 2 on: no taper, no source), advances the receiver field with damping, injects the time-reversed
    trace samples on column gz of the interior rows and adds F_k * r^{k+1} to the image.  Injection and imaging are pointwise in x, so
    they are local to whichever slab holds the row (ghost rows recompute them redundantly, as they recompute the fields); only the four
    fields travel.  Between two exchanges the valid rows shrink by h per iteration on the interior sides, exactly as in SlabForward."""

    def __init__(self, geom, stepper, snaps, rcv, v2, samples, gz, img, nt, group=None, overlap=True):
        """snaps = (P, PP) of the forward pass on this slab's rows (P = u^{nt-1} damped, PP = u^{nt}: fd-code.cu:502-507); rcv = two zero
        fields; samples = the shot gather transposed to [nt][nx] with row it = d_obs[.][nt-1-it] (what iteration it injects);
        img = [nxl][pitch] accumulator (only owned rows are meaningful afterwards)."""
        self.stepper = stepper
        self.f1, self.f0 = snaps[0], snaps[1]       # F_{k-1} (newer in backward time), F_{k-2}
        self.rn, self.ro = rcv[0], rcv[1]           # r^k (d_pr), r^{k-1} (d_ppr)
        self.v2, self.samples, self.gz, self.img, self.nt = v2, samples, gz, img, nt
        self.it = 0
        self._xfields = None
        self._init_loop(geom, group, overlap, self.f1.is_cuda, False)

    def exchange_fields(self):
        return self._xfields if self._xfields is not None else (self.f1, self.f0, self.rn, self.ro)

    def _roles_after(self):
        """(F_{k-1}, F_{k-2}, r^k, r^{k-1}) as the NEXT iteration sees them."""
        f1, f0 = (self.f0, self.f1) if self.it >= 2 else (self.f1, self.f0)      # F_k was written over F_{k-2}
        return (f1, f0, self.ro, self.rn)                                        # fd-code.cu:331-333

    def _iter(self, r0, r1, stream):
        if r1 <= r0:
            return
        if self.it < 2:        # the source field is a snapshot as it stands: iteration 0 images u^nt, iteration 1 u^{nt-1}
            F = self.f0 if self.it == 0 else self.f1
            self.stepper.back_iter(False, F, F, self.rn, self.ro, self.v2, r0, r1, self.it, self.samples, self.gz, self.img, stream)
        else:
            self.stepper.back_iter(True, self.f1, self.f0, self.rn, self.ro, self.v2, r0, r1, self.it, self.samples, self.gz, self.img, stream)

    def _advance(self):
        self.f1, self.f0, self.rn, self.ro = self._roles_after()
        self.it += 1

    def cycle(self, kk, more_after, stream):
        g = self.g
        yield "pre"
        split_last = self.overlap and g.world > 1 and kk == g.ksteps and more_after
        for j in range(1, kk + 1):
            r0, r1 = g.update_range(j)
            if split_last and j == kk:
                lo_end = r0 + g.G if g.has_lo else r0
                hi_beg = r1 - g.G if g.has_hi else r1
                self._iter(r0, lo_end, stream)
                self._iter(hi_beg, r1, stream)
                self._xfields = self._roles_after()  # what the exchange started at "mid" sends and fills: the next cycle's roles
                yield "mid"
                self._xfields = None
                self._iter(lo_end, hi_beg, stream)   # the interior rows of the same iteration, beside the transfer
                self._advance()
            else:
                self._iter(r0, r1, stream)
                self._advance()
