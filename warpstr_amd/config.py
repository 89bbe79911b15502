"""YAML configuration with the reference's keys and defaults (src/config.py:10-59, 91-171, src/default.yaml).

What steps 3 and 4 read is interpreted and honoured -- `output`, `loci`, `flank_length`, `threads`, `verbose`, `force_overwrite`,
`pore_model_path`, `tr_region_calling`, `genotyping`, `tr_calling_config`, `rescaling`, `genotyping_config` --; keys of the other
steps are preserved (`raw`) so that a WarpSTR config file can be passed unchanged; a setting that asks for something this package
does not do is REFUSED when it changes results (`genotyping_config.msa: True`) and reported once when it only adds a picture
(`visualize*`: the plots are upstream's).
"""
import copy
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import yaml

from .caller import CallerConfig, RescalerConfig

DEFAULTS: Dict[str, Any] = {   # src/default.yaml
    'verbose': 0, 'threads': 2, 'force_overwrite': False, 'flank_length': 110,
    'pore_model_path': 'example/deps/template_median68pA.model',
    'tr_calling_config': {'spike_removal': 'Brute', 'min_values_per_state': 4, 'states_in_segment': 6,
                          'min_state_similarity': 0.75, 'visualize_alignment': True, 'visualize_phase': True,
                          'visualize_strand': True, 'visualize_cost': True},
    'rescaling': {'reps_as_one': False, 'threshold': 0.5, 'max_std': 0.5, 'method': 'mean'},
    'genotyping_config': {'min_weight': 0.2, 'std_filter': 2, 'visualize': True, 'msa': False},
}
UPSTREAM_DEFAULT_PORE_MODEL = 'template_median68pA.model'   # the table this package carries as data (pore_model.py)


def add_defaults(config: Dict[str, Any], default: Dict[str, Any]) -> None:
    """Recursive defaults merge (src/config.py:31-59)."""
    for key, val in default.items():
        if isinstance(val, dict):
            config.setdefault(key, {})
            add_defaults(config[key], val)
        elif key not in config:
            config[key] = copy.deepcopy(val)


@dataclass
class GenotypingConfig:
    """src/config.py:122-131."""
    min_weight: float = 0.2
    std_filter: float = 2
    visualize: bool = True
    msa: bool = False

    def __post_init__(self):
        assert self.min_weight > 0 and self.min_weight < 1
        assert self.std_filter > 1
        if self.msa:
            raise ValueError('genotyping_config.msa: True asks for the MUSCLE alignment of the called sequences '
                             "(src/genotyper/muscle.py), which is upstream's and not part of this package: set it to False")

    def settings(self) -> Dict[str, float]:
        """What genotyper.run_genotyping_overview / run_genotyping_complex take."""
        return {'min_weight': float(self.min_weight), 'std_filter': float(self.std_filter)}


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
    genotyping: bool = False
    genotyping_config: GenotypingConfig = field(default_factory=GenotypingConfig)
    pore_model_path: Optional[str] = None
    force_overwrite: bool = False
    verbose: int = 0
    raw: Dict[str, Any] = field(default_factory=dict)

    def pore_model(self):
        """The PoreModel of `pore_model_path` (src/squiggler/pore_model.py:15-33).  Upstream's default path is relative to its
        own checkout; when that very file is not there, the same table is taken from this package's data.  Any other path
        that does not exist is upstream's error."""
        from .pore_model import PoreModel, default_pore_model
        path = self.pore_model_path
        if path is None:
            return default_pore_model()
        if not os.path.exists(path) and os.path.basename(path) == UPSTREAM_DEFAULT_PORE_MODEL:
            return default_pore_model()
        return PoreModel(path)

    def notices(self) -> List[str]:
        """One line per setting that is accepted and has no effect here (pictures)."""
        out = []
        vis = [k for k in ('visualize_alignment', 'visualize_phase', 'visualize_strand', 'visualize_cost')
               if self.raw.get('tr_calling_config', {}).get(k)]
        if self.tr_region_calling and vis:
            out.append(f"tr_calling_config.{{{', '.join(vis)}}}: the plots of step 3 are upstream's (src/caller/plotter.py) and not produced")
        if self.genotyping and self.genotyping_config.visualize:
            out.append("genotyping_config.visualize: summaries/alleles.svg is upstream's plot and not produced")
        return out


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
                         loci=loci, tr_region_calling=bool(cfg.get('tr_region_calling', True)), genotyping=bool(cfg.get('genotyping', False)),
                         genotyping_config=GenotypingConfig(**cfg['genotyping_config']), pore_model_path=cfg.get('pore_model_path'),
                         force_overwrite=bool(cfg['force_overwrite']), verbose=int(cfg['verbose'] or 0), raw=cfg)
