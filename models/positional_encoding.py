"""models.positional_encoding -- same public names as the reference's module."""
from vitpe.positional_encoding import (AbsolutePositionalEncoding, NoPositionalEncoding, PolynomialRPE,  # noqa: F401
                                       RelativePositionalEncoding, RoPEAxial, RoPEMixed)
