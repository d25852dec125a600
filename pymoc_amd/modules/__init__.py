from .column import Column
from .psi_thermwind import Psi_Thermwind
