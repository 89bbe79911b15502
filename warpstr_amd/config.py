"""YAML configuration with the reference's keys and defaults (src/config.py:10-59, src/default.yaml).

Only what step 3 needs is interpreted; unknown keys are preserved so that a WarpSTR config file can be
passed unchanged.
"""
import copy
from dataclasses import dataclass, field
from typing import Any, Dict, List

import yaml

from .caller import CallerConfig, RescalerConfig

DEFAULTS: Dict[str, Any] = {
    'verbose': 0, 'threads': 2, 'force_overwrite': False, 'flank_length': 110,
    'pore_model_path': None,
    'tr_calling_config': {'spike_removal': 'Brute', 'min_values_per_state': 4, 'states_in_segment': 6,
                          'min_state_similarity': 0.75, 'visualize_alignment': True, 'visualize_phase': True,
                          'visualize_strand': True, 'visualize_cost': True},
    'rescaling': {'reps_as_one': False, 'threshold': 0.5, 'max_std': 0.5, 'method': 'mean'},
}


def add_defaults(config: Dict[str, Any], default: Dict[str, Any]) -> None:
    """Recursive defaults merge (src/config.py:31-59)."""
    for key, val in default.items():
        if isinstance(val, dict):
            config.setdefault(key, {})
            add_defaults(config[key], val)
        elif key not in config:
            config[key] = copy.deepcopy(val)


@dataclass
class LocusConfig:
    name: str
    sequence: str
    flank_length: int
    coord: str = ''
    motif: str = ''


@dataclass
class WarpstrConfig:
    output: str
    threads: int
    flank_length: int
    caller: CallerConfig
    rescaler: RescalerConfig
    loci: List[LocusConfig] = field(default_factory=list)
    tr_region_calling: bool = True
    raw: Dict[str, Any] = field(default_factory=dict)


def load_config(path: str) -> WarpstrConfig:
    with open(path, 'r') as f:
        cfg = yaml.safe_load(f)
    if cfg is None:
        raise ValueError(f'Error when loading config file from {path}')
    add_defaults(cfg, DEFAULTS)
    if 'loci' not in cfg:
        raise KeyError('No loci defined in the config')
    loci = []
    for item in cfg['loci']:
        if not item.get('sequence'):
            raise ValueError(f"locus {item.get('name')}: only explicit `sequence` is supported by the caller step "
                             '(deriving it from `motif` needs the reference FASTA, src/schemas/locus.py:48-100)')
        loci.append(LocusConfig(name=item['name'], sequence=str(item['sequence']).upper(),
                                flank_length=int(item.get('flank_length') or cfg['flank_length']),
                                coord=item.get('coord', ''), motif=item.get('motif', '') or ''))
    return WarpstrConfig(output=cfg.get('output', '.'), threads=int(cfg['threads']), flank_length=int(cfg['flank_length']),
                         caller=CallerConfig(**cfg['tr_calling_config']), rescaler=RescalerConfig(**cfg['rescaling']),
                         loci=loci, tr_region_calling=bool(cfg.get('tr_region_calling', True)), raw=cfg)
