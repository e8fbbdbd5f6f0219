"""Command line -- same flags and report text as ``python -m open_pcc_metric`` (handler.py:4-71).

    python -m open_pcc_metric_amd --ocloud A.ply --pcloud B.ply [--pcloud C.ply ...] [--color rgb|ycc] [--hausdorff]
                                  [--point-to-plane] [--csv]

Extra, optional flags (defaults reproduce the reference): ``--device``, ``--engine``,
``--normal-index row|neighbour`` (row = the reference's D2, which raises IndexError when the clouds
differ in size), ``--extent X Y Z`` (inject the PSNR peak box instead of the
Qhull minimal-OBB restatement), ``--tie-exposure`` (diagnostic on stderr: how much of the point-to-plane result hangs
on the order of exact ties, which nanoflann decides by traversal -- cloud_pair.py:22-23).  Files without normals get them estimated on the GPU when
--point-to-plane asks for them (k = 30 covariance normals, as Open3D's estimate_normals does at
cloud_pair.py:61-64).  Input files: ply, pcd, xyz, xyzn, xyzrgb, pts (io.py; the formats
``o3d.io.read_point_cloud`` picks by extension, handler.py:57).
"""
import click


@click.command()
@click.option("--ocloud", required=True, type=str, help="Original point cloud.")
@click.option("--pcloud", required=True, type=str, multiple=True,
              help="Processed point cloud.  May be given several times: one report per processed cloud, printed one after the other "
                   "exactly as separate runs would print them, with the original cloud read, uploaded and analysed once.")
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
@click.option("--tie-exposure", is_flag=True,
              help="After the report, print to stderr how many points have several equidistant nearest neighbours and the "
                   "interval of point-to-plane MSE values the order of those ties can produce (diagnostic).")
def cli(ocloud, pcloud, color, hausdorff, point_to_plane, csv, device, engine, normal_index, extent, tie_exposure) -> None:
    from .calculator import MetricCalculator
    from .cloud_pair import CloudPair
    from .io import read_point_cloud
    from .options import CalculateOptions, transform_options

    ocloud_cloud = read_point_cloud(ocloud)
    options = CalculateOptions(color=color, hausdorff=hausdorff, point_to_plane=point_to_plane)
    cloud_pair = None
    for path in pcloud:
        pcloud_cloud = read_point_cloud(path)
        if cloud_pair is None:
            # (clouds read from files are freed while the GPU context works on -- with several decoded clouds, when the next one
            # is read --: their bytes go through the context's own pinned buffers, see CloudPair's staged_io; 0.6 ms for a pair)
            cloud_pair = CloudPair(ocloud_cloud, pcloud_cloud, device=device, nn_engine=engine, normal_index=normal_index,
                                   extent=list(extent) if extent else None, staged_io=True)
        else:
            cloud_pair = cloud_pair.with_reconst(pcloud_cloud)     # the original cloud stays in HBM with all that belongs to it
        calculator = MetricCalculator(cloud_pair)
        result = calculator.calculate(transform_options(options)).as_df()
        print(result.to_csv() if csv else result.to_string())
        if tie_exposure:                       # stderr: stdout stays byte-identical to the reference's report
            for is_left in (True, False):
                t = cloud_pair.tie_exposure(is_left, point_to_plane)
                line = (f"tie exposure ({t['direction']}): {t['tied_queries']} of {t['queries']} points have several equidistant nearest "
                        f"neighbours ({100.0 * t['tie_rate']:.3f} %, up to {t['max_multiplicity']})")
                if point_to_plane:
                    line += (f"; point-to-plane mse in [{t['d2_mse_min']!r}, {t['d2_mse_max']!r}] over all tie orders, "
                             f"reported {t['d2_mse_pick']!r} (smallest row)")
                click.echo(line, err=True)
    if cloud_pair is not None:
        cloud_pair.close()
