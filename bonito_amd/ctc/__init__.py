from .model import Model
from .basecall import basecall
