from .coerce import make_func, make_array
