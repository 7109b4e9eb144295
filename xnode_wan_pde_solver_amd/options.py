"""EngineOptions -- every switch of the engine and of train()'s loops in ONE object.

The process environment (XW_* variables) is read in exactly one place, `EngineOptions.from_env()`, once, when a solver (or a
bare Engine) is built; the object is handed down (NODE_WAN_solver.options -> Engine.options), printed by `solver.plan()`, and
tests / tools set its FIELDS (`EngineOptions(use_graphs=False)`, `dataclasses.replace(opts, xproj_min_d=1)`) instead of
patching the environment of the process.  The defaults are the measured optima of the headline workload (the comments in
engine.py next to where each field is used say where they come from)."""
import dataclasses
import os
from typing import Optional


def _flag(name, default):
    """'1' / '0' switch: anything but the other value keeps the default"""
    v = os.environ.get(name)
    if v is None:
        return default
    return v != '0' if default else v == '1'


def _int(name, default):
    v = os.environ.get(name)
    return default if v in (None, '') else int(v)


@dataclasses.dataclass
class EngineOptions:
    # ---- engine: scheduling -----------------------------------------------------------------------------------------------
    use_streams: bool = True            # XW_STREAMS: independent kernel chains of a sub-step on side streams
    use_graphs: bool = True             # XW_GRAPHS: capture each sub-step into a HIP graph and replay it
    strict_graphs: bool = False         # XW_STRICT_GRAPHS: a refused capture raises instead of falling back to eager launches
    reuse_test_net: bool = False        # XW_REUSE_V: v, dv/dt, nabla_x v(t_0) reused while phi and the sample are unchanged
    keep_activations: bool = True       # XW_KEEP_ACT: both forwards store their layer inputs for their backwards
    v_blocks: int = 0                   # XW_V_BLOCKS: block cap of the test network in a generator sub-step (0: 12/16 of the slots)
    v_blocks_disc: int = 0              # XW_V_BLOCKS_DISC: ... in a discriminator sub-step (0: 13/16)
    narrow: str = '1'                   # XW_NARROW: narrow stepper tiles -- 0 off, 1 auto, 2 wherever possible
    narrow_set: str = 'fx'              # XW_NARROW_SET: which launches may go narrow (f forward, x x-only sweep, p sweep with weight gradients)
    narrow_tiles: Optional[str] = None  # XW_NARROW_TILES: "f:x:p" largest launches (16-path tiles) that still go narrow
    narrow_tiles_alone: str = '256:256:64'   # XW_NARROW_TILES_ALONE: the same for a launch that has the chip to itself (test network reused)
    early_slab_sum: bool = True         # XW_EARLY_SLAB_SUM
    compact_tiles: int = 320            # XW_COMPACT_TILES: groups up to this many tiles take the compact generator schedule
    prio_drop_A: Optional[int] = None   # XW_PRIO_DROP_A: wave-priority drop of the generator's sweeps A + boundary (None: 3 up to d = 32, 2 above)
    prio_drop_X: int = 0                # XW_PRIO_DROP_X: ... of the discriminator's x-only sweep
    prio_drop_F: int = 0                # XW_PRIO_DROP_F: ... of the discriminator's forward pass
    prio_drop_G: int = 0                # XW_PRIO_DROP_G: ... of the generator's forward pass
    use_runner: bool = True             # XW_RUNNER: one C call per eager group sub-step (xw_substep_*)
    capture_exchange: bool = True       # XW_CAPTURE_EXCHANGE: several GPUs on RCCL -- the exchange inside the sub-step graphs
    xproj_min_d: int = 45               # XW_XPROJ_MIN_D: the test network's input layer once per path from this d on
    hw_queues: int = 4                  # GPU_MAX_HW_QUEUES as the HIP runtime sees it (a warning above 4: the schedule is laid out for 4)
    # ---- engine: semantics / checks ---------------------------------------------------------------------------------------
    verify_structure: bool = True       # XW_VERIFY_STRUCTURE: re-probe the coefficient structure every few samples
    packed_load: bool = True            # XW_PACKED_LOAD: list domains -- one upload + one gather launch per sample
    pairwise_single_slice: bool = True  # XW_ELEMENTWISE_SINGLE_SLICE=1 turns it off: the reference's [N, N] tables on single-slice groups
    adam_skips_untouched: bool = True   # XW_ADAM_NO_SKIP=1 turns it off: Adam skips parameters that got no gradient
    always_check: bool = False          # XW_ALWAYS_CHECK: status checks on every eager launch, not only the first 256
    poison: bool = False                # XW_POISON: work buffers start as NaN (tests: nothing reads what it did not write)
    # ---- train() loops (NODE_WAN_solver) ------------------------------------------------------------------------------------
    capture_refill: bool = True         # XW_CAPTURE_REFILL: refill of the cube group / diagnostic as one graph replay each
    defer_list_readback: bool = True    # XW_DEFER_LIST: list domains -- one read-back per outer iteration
    sampler_process: bool = True        # XW_SAMPLER_PROCESS: ball domains -- samples drawn by a forked child process
    show_plan: bool = False             # XW_SHOW_PLAN: train() prints solver.plan() once per call
    # ---- several GPUs (dist.World) ------------------------------------------------------------------------------------------
    replicate_below: int = 16           # XW_REPLICATE_BELOW: groups with fewer paths per rank are computed whole on every rank
    native_allreduce: bool = True       # XW_NATIVE_ALLREDUCE: the exchange through xw_allreduce (RCCL) instead of torch.distributed

    @classmethod
    def from_env(cls):
        """the one place where the XW_* environment variables are read"""
        o = cls()
        o.use_streams = _flag('XW_STREAMS', o.use_streams)
        o.use_graphs = _flag('XW_GRAPHS', o.use_graphs)
        o.strict_graphs = _flag('XW_STRICT_GRAPHS', o.strict_graphs)
        o.reuse_test_net = _flag('XW_REUSE_V', o.reuse_test_net)
        o.keep_activations = _flag('XW_KEEP_ACT', o.keep_activations)
        o.v_blocks = _int('XW_V_BLOCKS', o.v_blocks)
        o.v_blocks_disc = _int('XW_V_BLOCKS_DISC', o.v_blocks_disc)
        o.narrow = os.environ.get('XW_NARROW', o.narrow)
        o.narrow_set = os.environ.get('XW_NARROW_SET', o.narrow_set)
        o.narrow_tiles = os.environ.get('XW_NARROW_TILES') or None
        o.narrow_tiles_alone = os.environ.get('XW_NARROW_TILES_ALONE') or o.narrow_tiles_alone
        o.early_slab_sum = _flag('XW_EARLY_SLAB_SUM', o.early_slab_sum)
        o.compact_tiles = _int('XW_COMPACT_TILES', o.compact_tiles)
        o.prio_drop_A = _int('XW_PRIO_DROP_A', None)
        o.prio_drop_X = _int('XW_PRIO_DROP_X', o.prio_drop_X)
        o.prio_drop_F = _int('XW_PRIO_DROP_F', o.prio_drop_F)
        o.prio_drop_G = _int('XW_PRIO_DROP_G', o.prio_drop_G)
        o.use_runner = _flag('XW_RUNNER', o.use_runner)
        o.capture_exchange = _flag('XW_CAPTURE_EXCHANGE', o.capture_exchange)
        o.xproj_min_d = _int('XW_XPROJ_MIN_D', o.xproj_min_d)
        try:
            o.hw_queues = _int('GPU_MAX_HW_QUEUES', o.hw_queues)
        except ValueError:
            pass
        o.verify_structure = _flag('XW_VERIFY_STRUCTURE', o.verify_structure)
        o.packed_load = _flag('XW_PACKED_LOAD', o.packed_load)
        o.pairwise_single_slice = not _flag('XW_ELEMENTWISE_SINGLE_SLICE', False)
        o.adam_skips_untouched = not _flag('XW_ADAM_NO_SKIP', False)
        o.always_check = _flag('XW_ALWAYS_CHECK', o.always_check)
        o.poison = _flag('XW_POISON', o.poison)
        o.capture_refill = _flag('XW_CAPTURE_REFILL', o.capture_refill)
        o.defer_list_readback = _flag('XW_DEFER_LIST', o.defer_list_readback)
        o.sampler_process = _flag('XW_SAMPLER_PROCESS', o.sampler_process)
        o.show_plan = _flag('XW_SHOW_PLAN', o.show_plan)
        o.replicate_below = _int('XW_REPLICATE_BELOW', o.replicate_below)
        o.native_allreduce = _flag('XW_NATIVE_ALLREDUCE', o.native_allreduce)
        return o

    def non_default(self):
        """{field: value} of what differs from the defaults (solver.plan() prints it)"""
        base = EngineOptions()
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if getattr(self, f.name) != getattr(base, f.name)}
