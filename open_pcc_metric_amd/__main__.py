from . import handler

if __name__ == "__main__":
    # pylint: disable-next=no-value-for-parameter
    handler.cli()
