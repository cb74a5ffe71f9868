"""ctypes binding of the C-ABI in include/fmj.h (libfmj_hip.so, built in-tree by :func:`build`).

There is no CPU fallback: if the shared library is missing or no GPU is visible the product path
raises.  Loading the library and looking up its symbols does not need a GPU.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
SO_PATH = os.environ.get('FMJ_SO') or os.path.join(CSRC, 'libfmj_hip.so')      # FMJ_SO: A/B a variant build (scripts/)
HEADER = os.path.join(_HERE, '..', 'include', 'fmj.h')

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int32)
_D = ctypes.POINTER(ctypes.c_double)
_VP = ctypes.c_void_p


class FmjError(RuntimeError):
    """Non-zero status from the C-ABI (message from fmj_last_error)."""


class CData(ctypes.Structure):
    _fields_ = [(n, _VP) for n in ('qpos', 'qvel', 'ctrl', 'qpos_spring', 'xfrc_applied', 'xpos', 'xquat',
                                   'xipos', 'sensordata', 'qacc', 'time', 'status', 'qacc_warmstart', 'contact',
                                   'ncon')]


class CSensorLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('nsensordata', 'framelinvel_adr', 'jointpos_adr',
                                              'actuatorfrc_adr', 'first_link_body', 'first_sensor_jnt')]


class CRows(ctypes.Structure):
    _fields_ = [(n, _VP) for n in ('links', 'joints', 'xfrc', 'contacts')]


class CUnits(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ('meters', 'newtons', 'torques', 'velocity', 'angular_velocity',
                                              'kilograms')]


class CWater(ctypes.Structure):
    _fields_ = [('surface', ctypes.c_float), ('density', ctypes.c_float), ('viscosity', ctypes.c_float),
                ('velocity', ctypes.c_float*3), ('gravity', ctypes.c_float), ('use_buoyancy', ctypes.c_int32)]


class CWave(ctypes.Structure):
    _fields_ = [('amplitude', _VP), ('phase_lag', _VP), ('env_phase', _VP), ('frequency', ctypes.c_float)]


class CFusedArgs(ctypes.Structure):
    _fields_ = [('n_steps', ctypes.c_int32), ('iteration0', ctypes.c_int32), ('buffer_size', ctypes.c_int32),
                ('do_readout', ctypes.c_int32), ('do_drag', ctypes.c_int32), ('controller', ctypes.c_int32),
                ('ctrl_step_stride', ctypes.c_int64), ('row_stride_links', ctypes.c_int64),
                ('row_stride_joints', ctypes.c_int64), ('row_stride_xfrc', ctypes.c_int64),
                ('row_stride_contacts', ctypes.c_int64),
                ('rows_base', CRows), ('water', CWater), ('units', CUnits), ('wave', CWave), ('ctrl_out', _VP),
                ('env_order', _VP), ('substeps', ctypes.c_int32), ('substep_links', ctypes.c_int32),
                ('n_iterations', ctypes.c_int32), ('rows_ahead', ctypes.c_int32)]


BEFORE_ROWS, BEFORE_LINKS_ONLY, BEFORE_CONTACTS, BEFORE_DRAG = 1, 2, 4, 8      # FMJ_BEFORE_* of include/fmj.h


# every symbol include/fmj.h declares: name -> (restype, argtypes)
class CCpgDesc(ctypes.Structure):
    _fields_ = [('n_osc', ctypes.c_int32), ('n_conn', ctypes.c_int32), ('nu', ctypes.c_int32),
                ('frequency', _D), ('rate', _D), ('amplitude', _D),
                ('conn_to', _I), ('conn_from', _I), ('conn_weight', _D), ('conn_bias', _D),
                ('out_a', _I), ('out_b', _I), ('out_gain', _D), ('out_offset', _D)]


SYMBOLS = {
    'fmj_create': (ctypes.c_int, [_VP, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_VP)]),
    'fmj_destroy': (None, [_VP]),
    'fmj_last_error': (ctypes.c_char_p, []),
    'fmj_abi_version': (ctypes.c_int, []),
    'fmj_get_sensor_layout': (ctypes.c_int, [_VP, ctypes.POINTER(CSensorLayout)]),
    'fmj_kernel_info': (ctypes.c_int, [_VP, _I, _I]),
    'fmj_set_swimming': (ctypes.c_int, [_VP, ctypes.c_int32, ctypes.c_int32, _I, _I, _I, _D, _D, _D, _D]),
    'fmj_set_actuator_forcerange': (ctypes.c_int, [_VP, ctypes.c_int32, _I, _D]),
    'fmj_drag_link': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, _VP, ctypes.c_int64, _VP, ctypes.c_int64, _D,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(CWater), _VP, _VP]),
    'fmj_set_readout_maps': (ctypes.c_int, [_VP, ctypes.c_int32, _I, ctypes.c_int32, _I]),
    'fmj_step': (ctypes.c_int, [_VP, ctypes.POINTER(CData), ctypes.c_int32, ctypes.c_int64, _VP]),
    'fmj_forward': (ctypes.c_int, [_VP, ctypes.POINTER(CData), ctypes.c_int32, _VP]),
    'fmj_forward_debug': (ctypes.c_int, [_VP, ctypes.POINTER(CData), ctypes.c_int32, _VP, _I, _VP, _VP]),
    'fmj_step_debug': (ctypes.c_int, [_VP, ctypes.POINTER(CData), _VP, _VP, _VP]),
    'fmj_constraint_info': (ctypes.c_int, [_VP, _I, _I, _I]),
    'fmj_solver_info': (ctypes.c_int, [_VP, _I, _I, _I]),
    'fmj_drag': (ctypes.c_int, [_VP, ctypes.POINTER(CRows), ctypes.POINTER(CWater), ctypes.POINTER(CUnits), _VP, _VP]),
    'fmj_physics2data': (ctypes.c_int, [_VP, ctypes.POINTER(CData), ctypes.POINTER(CRows), ctypes.POINTER(CUnits),
                                        ctypes.c_int32, _VP]),
    'fmj_set_contact_maps': (ctypes.c_int, [_VP, ctypes.c_int32, _I, ctypes.c_int32, _I]),
    'fmj_before_step': (ctypes.c_int, [_VP, ctypes.POINTER(CData), ctypes.POINTER(CRows), ctypes.POINTER(CWater), ctypes.POINTER(CUnits),
                                       ctypes.c_int32, _VP, _VP]),
    'fmj_contacts2data': (ctypes.c_int, [_VP, ctypes.POINTER(CData), ctypes.POINTER(CRows), ctypes.POINTER(CUnits), _VP]),
    'fmj_step_fused': (ctypes.c_int, [_VP, ctypes.POINTER(CData), ctypes.POINTER(CFusedArgs), _VP]),
    'fmj_sc': (ctypes.c_int, [ctypes.c_char_p]),
    'fmj_cpg_create': (ctypes.c_int, [ctypes.POINTER(CCpgDesc), ctypes.c_int32, ctypes.POINTER(_VP)]),
    'fmj_cpg_destroy': (None, [_VP]),
    'fmj_cpg_tape': (ctypes.c_int, [_VP, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, _VP, _VP, _VP, _VP, _VP, _VP]),
}

def build_id() -> str:
    """sha1 (16 hex digits) of the sources libfmj_hip.so is built from (csrc/*.hip, csrc/*.inc, include/fmj.h): stamps measurements
    (bench.py, profiles/latest_traffic.json) so that counters taken on another build of the kernels are not reported as current."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith('.hip') or f.endswith('.inc'):
            h.update(f.encode()); h.update(open(os.path.join(CSRC, f), 'rb').read())
    h.update(open(HEADER, 'rb').read())
    return h.hexdigest()[:16] + ('-devlink' if os.path.exists(SO_PATH + '.devlink') else '')


_lib = None
ABI_VERSION = 6         # FMJ_ABI_VERSION of include/fmj.h


def build(force: bool = False, verbose: bool = False, defines=(), out: str = None) -> str:
    """Compile csrc/fmj_hip.hip for gfx950 into csrc/libfmj_hip.so (hipcc cross-compiles without a GPU).
    ``defines`` / ``out`` build a variant next to it (scripts/stamps.py: ``-DFMJ_STAMPS``).
    Development shortcut: ``FMJ_DEV_MAXD=20,28`` recompiles only the host object and the step-kernel objects of those register
    row lengths and links them with the objects a previous build left in csrc/_obj (possibly stale: never for a commit)."""
    src = os.path.join(CSRC, 'fmj_hip.hip')
    target = SO_PATH if out is None else os.path.join(CSRC, out)
    deps = [src, HEADER] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.inc')]
    stale = not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)
    if force or stale:
        # one object per register row length (the step-kernel instantiations) + one for the host side, compiled in
        # parallel, then linked: the single translation unit took 2.3 min, this takes the time of the slowest object
        import concurrent.futures
        flags = ['--offload-arch=gfx950', '-O3', '-fno-slp-vectorize', '-mllvm', '-pragma-unroll-threshold=131072', '-fPIC'] + list(defines)
        objdir = os.path.join(CSRC, '_obj' if out is None else '_obj_' + os.path.splitext(out)[0])
        os.makedirs(objdir, exist_ok=True)
        dev = [int(x) for x in os.environ.get('FMJ_DEV_MAXD', '').split(',') if x.strip()]
        jobs = [(os.path.join(objdir, 'host.o'), [])] + [(os.path.join(objdir, f'k{n}.o'), [f'-DFMJ_TU_MAXD={n}'])
                                                         for n in range(4, 65, 4)]      # 36 .. 64: the unconstrained one-env kernel only
        if dev:
            todo = [j for j in jobs if not j[1] or int(j[1][0].split('=')[1]) in dev or not os.path.exists(j[0])]
        else:
            todo = jobs

        def cc(job):
            # compiled under a per-process name, then renamed into place: two builds at once (two ranks, two pytest workers) never
            # read each other's half-written objects
            obj, defs = job
            tmp = f'{obj}.{os.getpid()}.tmp'
            cmd = ['hipcc'] + flags + defs + ['-c', src, '-o', tmp]
            if verbose:
                print(' '.join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, obj)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as ex:
            list(ex.map(cc, todo))
        tmp_target = f'{target}.{os.getpid()}.tmp'
        cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + [j[0] for j in jobs] + ['-o', tmp_target]
        if verbose:
            print(' '.join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp_target, target)
        finally:
            if os.path.exists(tmp_target):
                os.remove(tmp_target)
        # a development link (FMJ_DEV_MAXD) may hold stale objects: it leaves a marker next to the library that build_id() reports,
        # so a measurement taken on it cannot pass for one of a clean build; a full build removes the marker
        marker = target + '.devlink'
        if dev:
            open(marker, 'w').write(','.join(map(str, dev)))
        elif os.path.exists(marker):
            os.remove(marker)
    return target


OPTIONAL_IN_AB_BASE = ('fmj_solver_info',)      # queries only (physics.py tolerates their absence under FMJ_SO)


def load():
    """dlopen the HIP library and bind every declared symbol. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FmjError(f'{SO_PATH} is missing: the HIP extension is not built. Run '
                       '`python -c "import __graft_entry__ as g; g.build()"` (needs hipcc). '
                       'There is no CPU fallback for the product path.')
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        except AttributeError:
            if os.environ.get('FMJ_SO') and name in OPTIONAL_IN_AB_BASE:      # scripts/ab.sh: an older build under today's host code
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if lib.fmj_abi_version() != ABI_VERSION:
        raise FmjError('libfmj_hip.so ABI version mismatch; rebuild')
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise FmjError(f'fmj status {rc}: {load().fmj_last_error().decode()}')


def sc(name: str) -> int:
    v = load().fmj_sc(name.encode())
    if v < 0:
        raise KeyError(name)
    return v
