from typing import Optional, Union
from torch import Tensor

OptTensor = Optional[Tensor]
Adj = Union[Tensor, "SparseTensor"]
