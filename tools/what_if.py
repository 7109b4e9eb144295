"""What would the cycle gain if one of its kernels were FREE?  (timing only: the dropped launch leaves its outputs stale)

Each variant removes ONE launch from the captured sub-step graphs of bench.py's headline run and reports ms per sub-step;
t(shipped) - t(variant) is an upper bound on what ANY rewrite of that kernel can return -- and on designs that move its work
elsewhere: "one sweep, two cotangents" (sweep A's chain carried by sweep B behind the test network) can gain at most what
dropping sweep A's interior job gains, since the merged sweep costs at least what sweep B costs now.

    python tools/what_if.py [variant ...]        variants: none sweepA sweepA+bdry sweepB fwd_gen xsweep fwd_disc rec testnet_gen
                                                           contract (k_weak_partials, all three sub-steps) cot_disc adam (all three) bdry"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = ['none', 'sweepA', 'sweepA+bdry', 'sweepB', 'fwd_gen', 'xsweep', 'fwd_disc', 'rec', 'testnet_gen', 'contract', 'cot_disc', 'adam',
            'bdry', 'none']

if os.environ.get('XW_WHAT_IF') is None:
    rows = []
    for v in (sys.argv[1:] or VARIANTS):
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, XW_WHAT_IF=v), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if not line:
            print(v, 'FAILED', out.stderr[-400:])
            continue
        d = json.loads(line[-1])
        rows.append((v, d['ms_per_step'], d['value']))
        print('%-12s %.4f ms per sub-step  %7.1f sub-steps/s' % rows[-1], flush=True)
    base = [r[1] for r in rows if r[0] == 'none']
    if base:
        b = sum(base) / len(base)
        print('\nshipped: %.4f ms per sub-step (mean of %d runs); free kernel -> gain per sub-step (per g,g,d cycle)' % (b, len(base)))
        for v, ms, _ in rows:
            if v != 'none':
                print('  %-12s %+6.1f us  (%+6.1f us, %.1f %%)' % (v, 1e3 * (b - ms), 3e3 * (b - ms), 100 * (b - ms) / b))
    sys.exit(0)

sys.path.insert(0, os.path.join(HERE, '..'))
from xnode_wan_pde_solver_amd import kernels as KN
what = os.environ['XW_WHAT_IF']
bwd, fwd, dbwd, dfwd = KN.ode_bwd_multi, KN.ode_fwd_multi, KN.disc_bwd, KN.disc_fwd


def ode_bwd_multi(jobs, *a, **kw):
    params, x = kw.get('want_params'), kw.get('want_x')
    if params and kw.get('x_cot_ones'):                                # generator: fused sweep A (interior) + boundary sweep
        if what == 'sweepA+bdry':
            return None
        if what == 'sweepA':
            rest = [j for j in jobs if j.get('gx') is None]
            return bwd(rest, *a, **dict(kw, want_x=False, x_cot_ones=False)) if rest else None
    elif params and what == 'sweepB' and len(jobs) == 1 and 'weak' in (jobs[0].get('res') or {}):
        return None
    elif x and not params and what == 'xsweep':
        return None
    return bwd(jobs, *a, **kw)


def ode_fwd_multi(jobs, *a, **kw):
    if (what == 'fwd_disc' and kw.get('act_x_only')) or (what == 'fwd_gen' and not kw.get('act_x_only')):
        return None
    return fwd(jobs, *a, **kw)


def disc_bwd(*a, **kw):
    return None if what == 'rec' else dbwd(*a, **kw)


def disc_fwd(*a, **kw):
    return None if (what == 'testnet_gen' and kw.get('act') is None) else dfwd(*a, **kw)


KN.ode_bwd_multi, KN.ode_fwd_multi, KN.disc_bwd, KN.disc_fwd = ode_bwd_multi, ode_fwd_multi, disc_bwd, disc_fwd
for name, key in (('weak_partials', 'contract'), ('disc_cotangent', 'cot_disc'), ('adam', 'adam'), ('bdry_partials', 'bdry')):
    if what == key:
        setattr(KN, name, lambda *a, **kw: None)
import bench
sys.argv = ['bench.py', '--no-cpu-baseline', '--train-iters', '0', '--no-solo', '--steps', '90', '--warmup', '12'] + os.environ.get(
    'XW_WHAT_IF_ARGS', '').split()          # (e.g. XW_WHAT_IF_ARGS="--global-paths 512": a strong-scaling shard)
bench.main()
