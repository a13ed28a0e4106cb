"""Loss tags of the SVM dual path.

In the dual branch the reference uses its loss classes only as type tags (`self.loss == Hinge`,
optiml/ml/svm/_base.py:557, :727, :1102, :1279) and for `_loss_type` (losses.py:123, :195); the primal
objective math behind them is out of scope here.
"""

__all__ = ['Hinge', 'SquaredHinge', 'EpsilonInsensitive', 'SquaredEpsilonInsensitive',
           'hinge', 'squared_hinge', 'epsilon_insensitive', 'squared_epsilon_insensitive']


class SVMLoss:
    _loss_type = None

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('primal SVM objectives are outside the box-constrained dual path')


class Hinge(SVMLoss):
    _loss_type = 'classifier'


class SquaredHinge(SVMLoss):
    _loss_type = 'classifier'


class EpsilonInsensitive(SVMLoss):
    _loss_type = 'regressor'


class SquaredEpsilonInsensitive(SVMLoss):
    _loss_type = 'regressor'


hinge = Hinge
squared_hinge = SquaredHinge
epsilon_insensitive = EpsilonInsensitive
squared_epsilon_insensitive = SquaredEpsilonInsensitive
