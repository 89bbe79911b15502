"""`python -m warpstr_amd cfg.yaml` -- upstream's `python WarpSTR.py cfg.yaml` (WarpSTR.py:17-89) for the steps this package
implements: TR calling (step 3, on the GPU, every locus of the configuration through one handle) and genotyping (step 4).
The other steps of a configuration (read extraction, Guppy annotation, expected-signal generation, TR-region extraction) are
upstream's: their switches are reported and skipped, as a re-run with them set to False would skip them (README.md:169-180)."""
import sys

from .wrapper import main


def run(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 1 or argv[0].startswith('-'):
        raise SystemExit('usage: python -m warpstr_amd CONFIG.yaml [--genotype] [--segments-npz FILE]')
    import yaml
    with open(argv[0]) as f:
        cfg = yaml.safe_load(f) or {}
    others = [k for k in ('single_read_extraction', 'guppy_annotation', 'exp_signal_generation', 'tr_region_extraction') if cfg.get(k)]
    if others:
        print(f"warpstr_amd: steps {', '.join(others)} are upstream's (run them with WarpSTR.py); continuing with "
              'tr_region_calling / genotyping on their outputs', file=sys.stderr)
    return main(['--config', argv[0]] + argv[1:])


if __name__ == '__main__':
    run()
