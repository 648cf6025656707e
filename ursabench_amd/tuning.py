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


def use_shipped_miopen_db(prefix='ursa_miopen_'):
    """Call BEFORE the first convolution of the process. Respects an MIOPEN_USER_DB_PATH the caller already set;
    URSA_NO_SHIPPED_MIOPEN_DB=1 gives an empty private database instead (what round 2 ran with)."""
    if 'MIOPEN_USER_DB_PATH' in os.environ:
        return os.environ['MIOPEN_USER_DB_PATH']
    d = tempfile.mkdtemp(prefix=prefix)
    if os.environ.get('URSA_NO_SHIPPED_MIOPEN_DB') != '1' and os.path.isdir(SHIPPED):
        for f in os.listdir(SHIPPED):
            if f.endswith('.txt'):
                shutil.copy(os.path.join(SHIPPED, f), os.path.join(d, f))
    os.environ['MIOPEN_USER_DB_PATH'] = d
    return d
