"""The samples of a list domain (time-varying balls: 11-20 groups per sample, src/dataset.py:48-229) drawn in a PROCESS of their
own.

On those domains an outer iteration of train() is host work on both sides: the main thread queues ~60 sub-steps and loads the
next sample's groups (~2000 small tensor operations), the sampler builds two samples (the diagnostic's and the next iteration's).
As two THREADS they share the interpreter lock: alone the main thread needs 12 ms per outer iteration at config-5 size and the
sampler 11, together 20 (measured: tools/list_host_floor.py).  Here the sampler is a forked child (no exec; it never
touches the GPU) that owns both generator streams while train() runs -- the parent hands it torch's and numpy's global states at
the start of train() and takes them back at the end, so the streams end exactly where the reference's would -- and packs every
sample into a slot of shared memory that the parent has page-locked: the upload is one asynchronous copy straight from the slot.

Same draws, same order, same values as the sampling thread (solver._train.draw_ahead); tests/test_host_logic.py compares them.

ONE child per process, forked at the first train() that uses it and shared by every solver after that: a fork write-protects the
parent's pages, the GPU driver re-validates the process's memory on the next submission, and that next submission takes 0.15-0.35 s
in a small process and seconds in one that holds many allocations (measured: tools/sampler_proc_cost.py).  Sample slots are
attached by name, so a solver with larger samples gets larger slots without another fork.
"""
import atexit
import math
import os
import pickle
import traceback

import numpy as np
import torch

from . import sampling

_SERVER = None


@atexit.register
def _close_all():
    if _SERVER is not None:
        _SERVER.close()


SLOTS = 4          # two per request (diagnostic sample, next sample); a request reuses the slots of the one before the last


class PackedSample:
    """what the training loop reads of a sampling.Comb_loader, over a sample the child packed into a shared slot"""

    def __init__(self, buf, meta, guard=None):
        self._buf = buf
        self._total, self._offs, self._hints, self._n, self._nu = meta
        self._views = None
        self._guard = guard            # (slot events of the owning SamplerProcess: the upload below records one)

    def compact(self):
        return None

    def pin(self):
        return self

    def _host(self, lo, hi):
        return [self._buf[o:o + math.prod(shape)].view(shape) for o, shape in self._offs[lo:hi]]

    interioru = property(lambda self: self._host(0, self._nu))
    interiorv = property(lambda self: [g.clone() for g in self._host(0, self._nu)])
    boundary = property(lambda self: self._host(self._nu, len(self._offs)))

    def __len__(self):
        return self._nu

    def _device_views(self, device):
        if self._views is None:
            dev = self._buf[:self._total].to(device, non_blocking=True)
            if self._guard is not None:
                # the copy reads the shared slot asynchronously: the child may only overwrite the slot once this event has passed
                # (SamplerProcess.submit waits for it before it hands the slot out again)
                owner, slot = self._guard
                owner._slot_events[slot] = torch.cuda.current_stream(device).record_event()
            self._views = [dev[o:o + math.prod(shape)].view(shape) for o, shape in self._offs]
        return self._views

    def device_groups(self, device):
        v, n, nu = self._device_views(device), self._n, self._nu
        return [(v[k], v[k], v[nu + k]) for k in range(n)], self._hints[:n]

    def device_interior(self, device):
        return self._device_views(device)[:self._nu], self._hints


def _pack(loader, slot):
    """Comb_loader.pack into a shared slot; a sample that does not fit travels through the pipe instead (inline)"""
    total = sum(g.numel() for g in loader._cache['interioru']) + sum(g.numel() for g in loader._cache['boundary'])
    fits = total <= slot.numel()
    packed = loader.pack(into=slot if fits else None)
    if not packed:
        raise RuntimeError('the sample of this domain is not a list of float64 groups')
    _, buf, total, offs, hints, n, nu = packed
    return (total, offs, hints, n, nu), (None if fits else buf[:total].clone())


def _child(conn):
    """the sampling process: draws what solver._train.draw_ahead draws, in its order"""
    import _posixshmem
    import mmap
    try:
        import ctypes
        import signal
        ctypes.CDLL(None).prctl(1, int(signal.SIGKILL))      # PR_SET_PDEATHSIG: gone with the parent, however the parent goes
    except Exception:
        pass
    torch.set_num_threads(1)          # (a forked child must not enter an OpenMP region of the parent's thread pool)
    send = lambda obj: conn.send_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))                      # noqa: E731
    attached = {}
    domain = new_domain = slots = N_r = N_b = None
    try:
        while True:
            try:
                msg = pickle.loads(conn.recv_bytes())
            except EOFError:
                break
            try:
                if msg[0] == 'begin':
                    _, (domain_cls, setup, N_r, N_b), names, elems, t_state, n_state = msg
                    for name in names:
                        if name not in attached:
                            # (mapped by hand: multiprocessing's SharedMemory would start a resource tracker of this child's own,
                            #  which "cleans up" -- unlinks -- the parent's blocks when the child ends)
                            fd = _posixshmem.shm_open('/' + name, os.O_RDWR, mode=0o600)
                            try:
                                shm = mmap.mmap(fd, 8 * elems)
                            finally:
                                os.close(fd)
                            attached[name] = (shm, torch.frombuffer(shm, dtype=torch.float64, count=elems))
                    slots = [attached[name][1] for name in names]
                    new_domain = lambda: domain_cls(setup['shape_param'], setup['dim'], setup['T0'], setup['T'], setup['N_t'])   # noqa: E731
                    torch.set_rng_state(t_state)
                    np.random.set_state(n_state)
                    domain = None
                    send(('ok',))
                elif msg[0] == 'first':
                    domain = new_domain()
                    meta = _pack(sampling.Comb_loader(N_r, N_b, domain, 'cpu'), slots[msg[1]])
                    send(('ok', domain, meta))
                elif msg[0] == 'draw':
                    _, slot_a, slot_b, last = msg
                    after = _pack(sampling.Comb_loader(N_r, N_b, domain, 'cpu', interior_only=True), slots[slot_a])
                    if last:
                        send(('ok', after, None, None))
                    else:
                        domain = new_domain()
                        send(('ok', after, domain, _pack(sampling.Comb_loader(N_r, N_b, domain, 'cpu'), slots[slot_b])))
                elif msg[0] == 'finish':
                    send(('ok', torch.get_rng_state(), np.random.get_state()))
                elif msg[0] == 'drop':
                    for name in msg[1]:
                        ent = attached.pop(name, None)
                        if ent is not None:
                            slots = None
                            shm, t = ent
                            del t, ent
                            try:
                                shm.close()
                            except Exception:
                                pass
                    send(('ok',))
                elif msg[0] == 'quit':
                    break
            except Exception:
                send(('error', traceback.format_exc()))
    finally:
        os._exit(0)                   # (no interpreter shutdown: the parent's GPU runtime was copied into this process by the fork)


class _Server:
    """the one sampling child of this process and the shared, page-locked sample slots (kept per size)"""

    def __init__(self):
        import multiprocessing
        ctx = multiprocessing.get_context('fork')
        self.conn, child_conn = ctx.Pipe()
        self.proc = ctx.Process(target=_child, args=(child_conn,), daemon=True)
        self.proc.start()
        child_conn.close()
        self.pools = {}               # capacity in doubles -> the pool of SLOTS shared blocks (slots)
        self.serial = 0
        self.user = None              # the SamplerProcess that holds the streams right now

    def send(self, *msg):
        self.conn.send_bytes(pickle.dumps(msg, protocol=pickle.HIGHEST_PROTOCOL))

    def recv(self, timeout=600.0):
        """the child's answer.  The pipe is polled, not waited on blindly: a child that died (or hung -- it is a fork of a
        multi-threaded process) makes train() raise instead of waiting forever"""
        import time as _time
        t_end = _time.monotonic() + timeout
        try:
            while not self.conn.poll(0.25):
                if not self.proc.is_alive():
                    raise RuntimeError('the sampling process ended unexpectedly (exit code %r)' % self.proc.exitcode)
                if _time.monotonic() > t_end:
                    raise RuntimeError('the sampling process did not answer within %.0f s' % timeout)
            out = pickle.loads(self.conn.recv_bytes())
        except EOFError:
            raise RuntimeError('the sampling process ended unexpectedly (exit code %r)' % self.proc.exitcode)
        if out[0] == 'error':
            raise RuntimeError('the sampling process failed:\n' + out[1])
        return out[1:]

    def slots(self, elems):
        """-> (names, tensors, capacity) of the slot pool for samples of `elems` doubles.  The blocks are POSIX shared memory made
        and mapped by hand (multiprocessing.shared_memory would start its resource-tracker helper: an exec out of a process that
        holds the GPU) and unlinked as soon as the child has mapped them (unlink_attached): nothing is left in /dev/shm however
        this process ends."""
        import _posixshmem
        import mmap
        if not any(e >= elems for e in self.pools):
            st = os.statvfs('/dev/shm')
            if st.f_bavail * st.f_frsize < SLOTS * 8 * elems + (64 << 20):
                raise RuntimeError('not enough shared memory for %d sample slots of %d MB' % (SLOTS, 8 * elems >> 20))
            for old in list(self.pools):                          # (one pool is in use at a time: smaller ones go)
                self.drop(old)
            names, maps = [], []
            for k in range(SLOTS):
                name = 'xw_samples_%d_%d_%d' % (os.getpid(), self.serial, k)
                fd = _posixshmem.shm_open('/' + name, os.O_CREAT | os.O_EXCL | os.O_RDWR, mode=0o600)
                try:
                    os.ftruncate(fd, 8 * elems)
                    maps.append(mmap.mmap(fd, 8 * elems))
                finally:
                    os.close(fd)
                names.append(name)
            self.serial += 1
            tens = [torch.frombuffer(m, dtype=torch.float64, count=elems) for m in maps]
            locked = []
            if torch.cuda.is_available():
                rt = torch.cuda.cudart()
                for t in tens:
                    if int(rt.cudaHostRegister(t.data_ptr(), t.numel() * 8, 0)) == 0:
                        locked.append(t.data_ptr())
            self.pools[elems] = dict(names=names, maps=maps, tensors=tens, locked=locked, linked=True)
        usable = min(e for e in self.pools if e >= elems)
        pool = self.pools[usable]
        return pool['names'], pool['tensors'], usable

    def unlink_attached(self, elems):
        """the child has mapped the pool's blocks: their names can go (the mappings on both sides stay)"""
        import _posixshmem
        pool = self.pools.get(elems)
        if pool is not None and pool['linked']:
            pool['linked'] = False
            for name in pool['names']:
                try:
                    _posixshmem.shm_unlink('/' + name)
                except OSError:
                    pass

    def drop(self, elems):
        pool = self.pools.pop(elems)
        try:
            if self.proc.is_alive():
                self.send('drop', pool['names'])
                self.recv()
        except Exception:
            pass
        if pool['locked'] and torch.cuda.is_available():
            rt = torch.cuda.cudart()
            for p in pool['locked']:
                rt.cudaHostUnregister(p)
        if pool['linked']:
            self.pools[elems] = pool
            self.unlink_attached(elems)
            self.pools.pop(elems)
        pool['tensors'] = None
        for m in pool['maps']:
            try:
                m.close()             # (refused while a sample still views the slot: the mapping then goes with its last view)
            except Exception:
                pass

    def close(self):
        try:
            if self.proc.is_alive():
                self.send('quit')
                self.proc.join(timeout=5)
                if self.proc.is_alive():
                    self.proc.terminate()
        except Exception:
            pass
        for elems in list(self.pools):
            self.drop(elems)


def _server():
    global _SERVER
    if _SERVER is None or not _SERVER.proc.is_alive():
        if _SERVER is not None:
            _SERVER.close()
        _SERVER = _Server()
    return _SERVER


class _Future:
    def __init__(self, owner):
        self.owner = owner

    def result(self):
        return self.owner._take()


class SamplerProcess:
    """A solver's handle on the sampling child.  Stands where solver._train puts its one-thread pool:
    submit(draw_ahead, domain, last) -> future with result()."""

    def __init__(self, domain_cls, setup, N_r, N_b):
        d, L = setup['dim'], setup['N_t']
        self.elems = (N_r * (L + 1) + N_b * L) * (1 + d)   # every path at full length plus an entry point, a full shell per sample time
        self.config = (domain_cls, dict(setup), N_r, N_b)
        self.server = _server()
        self.server.slots(self.elems)                       # (fails here, not in train(), when there is no room for the slots)
        self.request = 0
        self.outstanding = None
        self.active = False
        self.slots = None
        self._slot_events = {}          # slot -> event of the last upload out of it (PackedSample._device_views)

    proc = property(lambda self: self.server.proc)

    def _sample(self, slot, packed):
        meta, inline = packed
        if inline is not None:
            return PackedSample(inline, meta)
        return PackedSample(self.slots[slot], meta, guard=(self, slot))

    def _release(self, *slots):
        """before the child is told to write into these slots: every upload that still reads them has finished"""
        for sl in slots:
            ev = self._slot_events.pop(sl, None)
            if ev is not None:
                ev.synchronize()

    def begin(self):
        """hand the generator streams over (they come back in shutdown)"""
        sv = self.server
        if sv.user is not None and sv.user is not self:
            raise RuntimeError('the sampling process is held by another solver that is training right now')
        names, self.slots, elems = sv.slots(self.elems)
        sv.send('begin', self.config, names, elems, torch.get_rng_state(), np.random.get_state())
        sv.recv()
        sv.unlink_attached(elems)
        sv.user = self
        self.active, self.outstanding = True, None

    def first(self):
        slot = (2 * self.request) % SLOTS
        self.request += 1
        self._release(slot)
        self.server.send('first', slot)
        domain, packed = self.server.recv()
        return domain, self._sample(slot, packed)

    def submit(self, _fn, _domain, last):
        a = (2 * self.request) % SLOTS
        self.request += 1
        self._release(a, a + 1)
        self.server.send('draw', a, a + 1, bool(last))
        self.outstanding = (a, a + 1)
        return _Future(self)

    def _take(self):
        a, b = self.outstanding
        self.outstanding = None
        after, domain, nxt = self.server.recv()
        return self._sample(a, after), domain, (self._sample(b, nxt) if nxt is not None else None)

    def shutdown(self, wait=True):
        """end of train(): the streams go back to this process's generators"""
        if not self.active:
            return
        self.active = False
        import sys
        import warnings
        failing = sys.exc_info()[0] is not None       # (called from a `finally` while train()'s own exception propagates)
        try:
            if self.outstanding is not None:      # (train() left early: the request in flight is drained first)
                self._take()
            self.server.send('finish')
            t_state, n_state = self.server.recv()
            torch.set_rng_state(t_state)
            np.random.set_state(n_state)
        except Exception as e:
            # the generator streams the child held are lost: torch's and numpy's global generators stay where they were when
            # train() began.  Never at the expense of the exception that is already on its way
            if not failing:
                raise
            warnings.warn('the sampling process could not hand the generator streams back (%s): torch / numpy global RNG states are '
                          'those from before train()' % e, RuntimeWarning)
        finally:
            self.server.user = None

    def close(self):
        """(the child and the slots belong to the process and stay for the next solver)"""
        if self.active:
            self.shutdown()
        self.slots = None
