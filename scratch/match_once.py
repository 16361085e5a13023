import sys; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth
import bench
