"""Drop-in `nnutils` surface of the reference for the hot path: softras, loss_utils
(chamfer), geom_utils (projection) and train_utils (trainer loop with DDP)."""
