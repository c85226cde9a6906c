"""Build the product model (mdf-net_amd/net) through the product's own config counterpart."""
import contextlib
import io


def build_model():
    with contextlib.redirect_stdout(io.StringIO()):
        import config
        return config.build_model()
