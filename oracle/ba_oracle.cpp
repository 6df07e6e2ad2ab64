// placeholder until the BA oracle lands
