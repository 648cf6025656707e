"""MIOpen user databases for the benchmark networks.

Model forward / backward stay stock PyTorch-ROCm (MIOpen). MIOpen picks a solver per layer shape by a quick search on
first use and caches the result in a *user database* under $HOME. `ursabench_amd/miopen_db/` holds the databases MIOpen
itself wrote during one EXHAUSTIVE search (`MIOPEN_FIND_ENFORCE=3`, tools/miopen_tune.sh) over the layer shapes of the
benchmark configurations; `use_shipped_miopen_db()` gives the process a private, writable copy of them, so every run
starts from the tuned choices instead of re-doing the quick search (+4.6 % posterior-samples/s on BASELINE configs[1],
DESIGN.md §6). Nothing here replaces a MIOpen kernel: the files only name which of MIOpen's own solvers to use. They
are keyed by MIOpen build and gfx950; on any other stack MIOpen ignores them and behaves as before.

A private copy per process also keeps the find-db hazard of DESIGN.md §6 away (a search recorded under other
process-wide switches, e.g. deterministic mode, being reused)."""
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
SHIPPED = os.path.join(_HERE, 'miopen_db')


_OWNER = 'ursa_owner.pid'


def _here():
    """Which machine AND pid namespace this process lives in: hostname, boot id, and the pid-namespace inode. `os.kill(pid, 0)`
    only says something about pids of THIS namespace on THIS host: with a temp directory shared between nodes or containers
    (cluster scratch, a bind-mounted /tmp) a rank elsewhere would find "no such pid" for a live owner and delete its MIOpen
    databases in the middle of a run. A marker written elsewhere is therefore never acted upon."""
    import socket
    parts = [socket.gethostname()]
    for path in ('/proc/sys/kernel/random/boot_id',):
        try:
            parts.append(open(path).read().strip())
        except OSError:
            parts.append('?')
    try:
        parts.append(os.readlink('/proc/self/ns/pid'))
    except OSError:
        parts.append('?')
    return '|'.join(parts)


def _sweep_dead_owners():
    """Every bench rank, experiment / time_script main and smoke() makes a private directory; long sweeps used to pile
    them up under the temp directory. A directory is removed by the NEXT process that comes through here once its owner
    (the pid in its marker file, written together with the host / boot / pid-namespace it is valid in) no longer runs HERE - not at the owner's own exit, where MIOpen may still be flushing its
    databases into it. Directories without a marker (made by something else) are never touched."""
    tmp = tempfile.gettempdir()
    try:
        names = os.listdir(tmp)
    except OSError:
        return
    for name in names:
        d = os.path.join(tmp, name)
        if not (name.startswith('ursa_') and 'miopen' in name and os.path.isdir(d)):
            continue
        try:
            pid_s, _, where = open(os.path.join(d, _OWNER)).read().strip().partition(' ')
            pid = int(pid_s)
        except (OSError, ValueError):
            continue
        if where != _here():                      # another host / container / boot (or an old-format marker): not ours to judge
            continue
        try:
            os.kill(pid, 0)                       # signal 0: existence check only
        except ProcessLookupError:
            shutil.rmtree(d, ignore_errors=True)
        except OSError:
            pass                                  # exists but not ours (EPERM): leave it


def use_shipped_miopen_db(prefix='ursa_miopen_'):
    """Call BEFORE the first convolution of the process. Respects an MIOPEN_USER_DB_PATH the caller already set;
    URSA_NO_SHIPPED_MIOPEN_DB=1 gives an empty private database instead (what round 2 ran with)."""
    if 'MIOPEN_USER_DB_PATH' in os.environ:
        return os.environ['MIOPEN_USER_DB_PATH']
    _sweep_dead_owners()
    d = tempfile.mkdtemp(prefix=prefix)
    with open(os.path.join(d, _OWNER), 'w') as f:
        f.write(f'{os.getpid()} {_here()}')
    if os.environ.get('URSA_NO_SHIPPED_MIOPEN_DB') != '1' and os.path.isdir(SHIPPED):
        for f in os.listdir(SHIPPED):
            if f.endswith('.txt'):
                shutil.copy(os.path.join(SHIPPED, f), os.path.join(d, f))
    os.environ['MIOPEN_USER_DB_PATH'] = d
    return d
