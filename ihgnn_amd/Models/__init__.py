"""Model zoo of the hypergraph path, with the reference's name-lookup tables (``Models/__init__.py:9-24``)."""
from typing import Dict, Optional, Union

from .CommonLayers import FeatureInteractor
from .EmbeddingLayers import EmbeddingLayer
from .GnnLayers import GATLayer, GCNLayer, HGCNLayer, IHGNNLayer
from .PredictionLayers import HemPredictionLayer
from .RawGnn import RawGnn

PpsModel = Union[RawGnn]
PpsModelTypes = [RawGnn]
GnnLayer = Union[GCNLayer, GATLayer, HGCNLayer, IHGNNLayer]
GnnLayerTypes = [GCNLayer, GATLayer, HGCNLayer, IHGNNLayer]


def _short(name: str) -> str:
    return name[:-len('Layer')] if name.endswith('Layer') else name


# accepted spellings: class name, lower / upper case of it; for layers also without the "Layer" suffix; '' = driver default
parse_model_type: Dict[str, Optional[type]] = {'': None}
for _t in PpsModelTypes:
    for _k in (_t.__name__, _t.__name__.lower(), _t.__name__.upper()):
        parse_model_type[_k] = _t

parse_gnn_layer: Dict[str, Optional[type]] = {'': None}
for _t in GnnLayerTypes:
    for _k in (_t.__name__, _short(_t.__name__), _short(_t.__name__).lower()):
        parse_gnn_layer[_k] = _t
