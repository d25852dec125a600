"""pymoc_amd -- MI355X (gfx950) engine for the PyMOC timestep() path.

Host code is Python + ctypes over the C-ABI of libpymoc_hip.so (include/pymoc_hip.h);
there is no CPU fallback and no PyTorch in the product path.
"""
__version__ = "0.1.0"

from . import _lib
from .device import DeviceArray, Stream, Event, Graph, synchronize
from .columns import ColumnBatch
from .thermwind import ThermwindBatch
from .psi_so import PsiSOBatch
from .so_ml import SOMLBatch
from .equilibrium import ColumnEquiBatch
from .equi_column import EquiColumnBatch
from . import modules
from . import utils
from .modules import Column, Psi_Thermwind, Psi_SO, SO_ML, Equi_Column
from . import configs
from . import sharding
from .ensembles import (EquiIterationEnsemble, ColumnThermwindEnsemble, TwoColEnsemble, JN2018Ensemble,
                        TwoBasinEnsemble)
from . import diagnostics
