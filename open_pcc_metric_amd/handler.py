"""Command line -- same flags and report text as ``python -m open_pcc_metric`` (handler.py:4-71).

    python -m open_pcc_metric_amd --ocloud A.ply --pcloud B.ply [--color rgb|ycc] [--hausdorff]
                                  [--point-to-plane] [--csv]

Extra, optional flags (defaults reproduce the reference): ``--device``, ``--engine``,
``--normal-index row|neighbour`` (row = the reference's D2, which raises IndexError when the clouds
differ in size), ``--extent X Y Z`` (inject the PSNR peak box instead of the
Qhull minimal-OBB restatement).  Files without normals get them estimated on the GPU when
--point-to-plane asks for them (k = 30 covariance normals, as Open3D's estimate_normals does at
cloud_pair.py:61-64).  Input files: ply, pcd, xyz, xyzn, xyzrgb, pts (io.py; the formats
``o3d.io.read_point_cloud`` picks by extension, handler.py:57).
"""
import click


@click.command()
@click.option("--ocloud", required=True, type=str, help="Original point cloud.")
@click.option("--pcloud", required=True, type=str, help="Processed point cloud.")
@click.option("--color", required=False, type=click.Choice(["rgb", "ycc"]), help="Report color distortions as well.")
@click.option("--hausdorff", required=False, is_flag=True,
              help="Report hausdorff metric as well. If --point-to-plane is provided, "
                   "then hausdorff point-to-plane would be reported too")
@click.option("--point-to-plane", required=False, is_flag=True, help="Report point-to-plane distance as well.")
@click.option("--csv", required=False, is_flag=True, help="Print output in csv format.")
@click.option("--device", type=int, default=0, show_default=True, help="GPU to use.")
@click.option("--engine", type=click.Choice(["auto", "grid", "brute"]), default="auto", show_default=True,
              help="Exact nearest-neighbour engine.")
@click.option("--normal-index", type=click.Choice(["row", "neighbour"]), default="row", show_default=True,
              help="Which normal the point-to-plane projection uses.")
@click.option("--extent", type=float, nargs=3, default=None, help="Extents of the PSNR peak box (skips the min-OBB).")
def cli(ocloud, pcloud, color, hausdorff, point_to_plane, csv, device, engine, normal_index, extent) -> None:
    from .calculator import MetricCalculator
    from .cloud_pair import CloudPair
    from .io import read_point_cloud
    from .options import CalculateOptions, transform_options

    ocloud_cloud, pcloud_cloud = map(read_point_cloud, (ocloud, pcloud))
    cloud_pair = CloudPair(ocloud_cloud, pcloud_cloud, device=device, nn_engine=engine, normal_index=normal_index,
                           extent=list(extent) if extent else None)
    calculator = MetricCalculator(cloud_pair)
    options = CalculateOptions(color=color, hausdorff=hausdorff, point_to_plane=point_to_plane)
    result = calculator.calculate(transform_options(options)).as_df()
    print(result.to_csv() if csv else result.to_string())
