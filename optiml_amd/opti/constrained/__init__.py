__all__ = ['BoxConstrainedQuadraticOptimizer', 'ProjectedGradient', 'ActiveSet', 'FrankWolfe', 'InteriorPoint']

from ._base import BoxConstrainedQuadraticOptimizer
from .projected_gradient import ProjectedGradient
from .active_set import ActiveSet
from .frank_wolfe import FrankWolfe
from .interior_point import InteriorPoint
