"""Grid context for a function over 2 dimensions -- mirror of the reference's `contourist/field2d.py`
(Function2DGrid :8-9 is a thin wrapper over grid_field.FunctionGrid)."""
from . import grid_field


def Function2DGrid(xmin, ymin, xmax, ymax, dx, dy, function, materialize=False, cache=False):
    return grid_field.FunctionGrid((xmin, ymin), (xmax, ymax), (dx, dy), function, materialize, cache)
