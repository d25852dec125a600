from .coerce import make_func, make_array, check_numpy_version
