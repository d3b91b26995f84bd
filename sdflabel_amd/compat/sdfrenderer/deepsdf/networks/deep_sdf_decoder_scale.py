from sdflabel_amd.deepsdf.networks.deep_sdf_decoder_scale import Decoder  # noqa: F401
