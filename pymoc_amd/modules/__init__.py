from .column import Column
from .psi_thermwind import Psi_Thermwind
from .psi_SO import Psi_SO
from .SO_ML import SO_ML
from .equi_column import Equi_Column
