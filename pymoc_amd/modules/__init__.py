from .column import Column
