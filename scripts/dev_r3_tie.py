import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.point_cloud import PointCloud
a, b = bench.synth_content()
with CloudPair(PointCloud(a), PointCloud(b), extent=[511.0, 322.0, 505.0], normal_index="neighbour") as pair:
    for is_left in (True, False):
        print(pair.tie_exposure(is_left, True))
