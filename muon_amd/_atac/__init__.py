from . import preproc as pp
from . import tools as tl
