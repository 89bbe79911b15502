"""Import the upstream WarpSTR caller (read-only at /root/reference) under stubs.

DEVELOPMENT-CONTAINER ONLY.  Nothing in the product, the gpu tests, smoke() or
bench.py imports this module: /root/reference does not exist on the GPU box.
It is used by generate_golden.py (to produce the committed fixtures) and by
the optional `reference`-marked tests that cross-check the oracle live.

Recipe follows SURVEY.md section 8c: numpy-2 alias for np.bool8, dummy modules
for the third-party packages that are absent here (h5py, pysam, seaborn, Bio),
cwd inside the reference (relative pore_model_path) and a minimal YAML config
passed through sys.argv because src/config.py parses argv at import time.
"""
import os
import sys
import tempfile
import types

REFERENCE_ROOT = '/root/reference'


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'src', 'caller'))


_COMPLEMENT = str.maketrans('ACGTacgt', 'TGCAtgca')


class _Seq:  # minimal Bio.Seq.Seq stand-in (only reverse_complement is used)
    def __init__(self, s):
        self._s = str(s)

    def reverse_complement(self):
        return _Seq(self._s.translate(_COMPLEMENT)[::-1])

    def __str__(self):
        return self._s


def import_reference(min_values_per_state=4, states_in_segment=6, threshold=0.5,
                     max_std=0.5, method='mean', reps_as_one=False, flank_length=110, pore_model_path=None, genotyping_config=None):
    """Returns a namespace with the reference's caller symbols.  One config per process."""
    if not reference_available():
        raise RuntimeError('reference not present')
    import numpy as np
    if not hasattr(np, 'bool8'):
        np.bool8 = np.bool_
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use('Agg')
    for name in ['h5py', 'pysam', 'seaborn', 'Bio', 'Bio.Seq', 'Bio.pairwise2', 'Bio.SeqIO',
                 'Bio.Align', 'Bio.Align.Applications']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['Bio.Seq'].Seq = _Seq
    sys.modules['Bio'].pairwise2 = sys.modules['Bio.pairwise2']
    sys.modules['Bio'].SeqIO = sys.modules['Bio.SeqIO']
    sys.modules['Bio.Align.Applications'].MuscleCommandline = object

    tmpdir = tempfile.mkdtemp(prefix='warpstr_ref_')
    cfg = os.path.join(tmpdir, 'cfg.yaml')
    with open(cfg, 'w') as f:
        f.write(f"""reference_path: none
output: {tmpdir}
single_read_extraction: False
guppy_annotation: False
exp_signal_generation: False
tr_region_extraction: False
tr_region_calling: True
genotyping: False
flank_length: {flank_length}
{('pore_model_path: ' + pore_model_path) if pore_model_path else ''}
{('genotyping_config: ' + repr(dict(genotyping_config))) if genotyping_config else ''}
tr_calling_config:
  min_values_per_state: {min_values_per_state}
  states_in_segment: {states_in_segment}
  visualize_alignment: False
  visualize_phase: False
  visualize_strand: False
  visualize_cost: False
rescaling:
  reps_as_one: {reps_as_one}
  threshold: {threshold}
  max_std: {max_std}
  method: {method}
loci:
  - name: X
    coord: chr1:1-2
    sequence: (AGC)
""")
    old_cwd, old_argv = os.getcwd(), sys.argv
    os.chdir(REFERENCE_ROOT)
    sys.path.insert(0, REFERENCE_ROOT)
    sys.argv = ['WarpSTR.py', cfg]
    try:
        from src.caller import automata, caller, wrapper
        from src.squiggler.pore_model import pore_model
        from src.schemas.fast5 import Fast5, normalize_signal_mad
        from src.schemas import ReadSignal
        import src.templates as templates
    finally:
        os.chdir(old_cwd)
        sys.argv = old_argv
        sys.path.remove(REFERENCE_ROOT)
    ns = types.SimpleNamespace(automata=automata, caller=caller, wrapper=wrapper, pore_model=pore_model,
                               normalize_signal_mad=normalize_signal_mad, Fast5=Fast5, ReadSignal=ReadSignal,
                               templates=templates)
    return ns
