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
"""
import math
import os
import pickle
import traceback

import numpy as np
import torch

from . import sampling

import atexit
import weakref

_LIVE = weakref.WeakSet()


@atexit.register
def _close_all():
    for sp in list(_LIVE):
        sp.close()


SLOTS = 4          # two per request (diagnostic sample, next sample); a request reuses the slots of the one before the last


class PackedSample:
    """what the training loop reads of a sampling.Comb_loader, over a sample the child packed into a shared slot"""

    def __init__(self, buf, meta):
        self._buf = buf
        self._total, self._offs, self._hints, self._n, self._nu = meta
        self._views = None

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


def _child(conn, slots, solver_bits):
    """the sampling process: draws what solver._train.draw_ahead draws, in its order"""
    domain_cls, setup, N_r, N_b = solver_bits
    try:
        import ctypes
        import signal
        ctypes.CDLL(None).prctl(1, int(signal.SIGKILL))      # PR_SET_PDEATHSIG: gone with the parent, however the parent goes
    except Exception:
        pass
    torch.set_num_threads(1)          # (a forked child must not enter an OpenMP region of the parent's thread pool)
    new_domain = lambda: domain_cls(setup['shape_param'], setup['dim'], setup['T0'], setup['T'], setup['N_t'])   # noqa: E731
    send = lambda obj: conn.send_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))                      # noqa: E731
    domain = None
    try:
        while True:
            try:
                msg = pickle.loads(conn.recv_bytes())
            except EOFError:
                break
            try:
                if msg[0] == 'begin':
                    torch.set_rng_state(msg[1])
                    np.random.set_state(msg[2])
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
                elif msg[0] == 'quit':
                    break
            except Exception:
                send(('error', traceback.format_exc()))
    finally:
        os._exit(0)                   # (no interpreter shutdown: the parent's GPU runtime was copied into this process by the fork)


class _Future:
    def __init__(self, owner):
        self.owner = owner

    def result(self):
        return self.owner._take()


class SamplerProcess:
    """Stands where solver._train puts its one-thread pool: submit(draw_ahead, domain, last) -> future with result()."""

    def __init__(self, domain_cls, setup, N_r, N_b):
        import multiprocessing
        from multiprocessing import shared_memory
        d, L = setup['dim'], setup['N_t']
        elems = (N_r * (L + 1) + N_b * L) * (1 + d)       # every path at full length plus an entry point, a full shell per sample time
        self.N_r, self.N_b = N_r, N_b
        st = os.statvfs('/dev/shm')
        if st.f_bavail * st.f_frsize < SLOTS * 8 * elems + (64 << 20):
            raise RuntimeError('not enough shared memory for %d sample slots of %d MB' % (SLOTS, 8 * elems >> 20))
        self.shm = [shared_memory.SharedMemory(create=True, size=8 * elems) for _ in range(SLOTS)]
        self.slots = [torch.frombuffer(s.buf, dtype=torch.float64, count=elems) for s in self.shm]
        ctx = multiprocessing.get_context('fork')
        self.conn, child_conn = ctx.Pipe()
        self.proc = ctx.Process(target=_child, args=(child_conn, self.slots, (domain_cls, setup, N_r, N_b)), daemon=True)
        self.proc.start()
        child_conn.close()
        self.registered = []
        if torch.cuda.is_available():                      # (after the fork: the child maps the plain shared pages)
            rt = torch.cuda.cudart()
            for t in self.slots:
                if int(rt.cudaHostRegister(t.data_ptr(), t.numel() * 8, 0)) == 0:
                    self.registered.append(t.data_ptr())
        self.request = 0
        self.outstanding = None
        self.active = False
        _LIVE.add(self)

    # -- protocol ------------------------------------------------------------------------------------------------------
    def _send(self, *msg):
        self.conn.send_bytes(pickle.dumps(msg, protocol=pickle.HIGHEST_PROTOCOL))

    def _recv(self):
        try:
            out = pickle.loads(self.conn.recv_bytes())
        except EOFError:
            raise RuntimeError('the sampling process ended unexpectedly (exit code %r)' % self.proc.exitcode)
        if out[0] == 'error':
            raise RuntimeError('the sampling process failed:\n' + out[1])
        return out[1:]

    def _sample(self, slot, packed):
        meta, inline = packed
        return PackedSample(self.slots[slot] if inline is None else inline, meta)

    def begin(self):
        """hand the generator streams over (they come back in shutdown)"""
        self._send('begin', torch.get_rng_state(), np.random.get_state())
        self._recv()
        self.active, self.outstanding = True, None

    def first(self):
        slot = (2 * self.request) % SLOTS
        self.request += 1
        self._send('first', slot)
        domain, packed = self._recv()
        return domain, self._sample(slot, packed)

    def submit(self, _fn, _domain, last):
        a = (2 * self.request) % SLOTS
        self.request += 1
        self._send('draw', a, a + 1, bool(last))
        self.outstanding = (a, a + 1)
        return _Future(self)

    def _take(self):
        a, b = self.outstanding
        self.outstanding = None
        after, domain, nxt = self._recv()
        return self._sample(a, after), domain, (self._sample(b, nxt) if nxt is not None else None)

    def shutdown(self, wait=True):
        """end of train(): the streams go back to this process's generators"""
        if not self.active:
            return
        self.active = False
        if self.outstanding is not None:      # (train() left early: the request in flight is drained first)
            self._take()
        self._send('finish')
        t_state, n_state = self._recv()
        torch.set_rng_state(t_state)
        np.random.set_state(n_state)

    def close(self):
        try:
            if self.proc.is_alive():
                self._send('quit')
                self.proc.join(timeout=5)
                if self.proc.is_alive():
                    self.proc.terminate()
        except Exception:
            pass
        if self.registered and torch.cuda.is_available():
            rt = torch.cuda.cudart()
            for p in self.registered:
                rt.cudaHostUnregister(p)
        self.registered, self.slots = [], []
        for s in self.shm:
            try:
                s.unlink()
            except Exception:
                pass
            try:
                s.close()             # (refused while a sample still views the slot: the mapping then goes with its last view)
            except Exception:
                pass
        self.shm = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
