# config-1 end-to-end leg of bench.py alone (bin/sfm_native on the fountain images): wall seconds and the driver's stage split
import sys, json; sys.path.insert(0, '.')
import bench
print(json.dumps(bench.e2e_leg('config1', 'S', 300, False), indent=1))
