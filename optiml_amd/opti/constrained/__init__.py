__all__ = ['BoxConstrainedQuadraticOptimizer', 'AugmentedLagrangianQuadratic', 'ProjectedGradient', 'ActiveSet',
           'FrankWolfe', 'InteriorPoint', 'ActiveSetCG']

from ._base import BoxConstrainedQuadraticOptimizer, AugmentedLagrangianQuadratic
from .projected_gradient import ProjectedGradient
from .active_set import ActiveSet, ActiveSetCG
from .frank_wolfe import FrankWolfe
from .interior_point import InteriorPoint
