// Configurations of the LDS-DMA weight-gradient kernel that are compiled in: X(BO, BI, WO, WI, D, KPS) = tile
// (output channels x input channels), wave grid, ring depth, pixel rows per ring stage.
#pragma once
#define LH_WGRAD_CFGS(X) \
    X(256,256,2,4,3,32) X(256,256,2,4,4,32) X(256,256,2,4,2,64) X(128,128,2,2,2,32) \
    X(128,128,2,2,4,32) X(128,128,2,2,2,64) X(128,128,2,2,3,64) X(128,64,4,1,2,32) \
    X(128,64,4,1,4,32) X(128,64,4,1,2,64) X(128,64,4,1,3,64) X(64,128,1,4,2,32) \
    X(64,128,1,4,4,32) X(64,128,1,4,2,64) X(64,128,1,4,3,64) X(64,64,2,2,2,32) \
    X(64,64,2,2,4,32) X(64,64,2,2,2,64) X(64,64,2,2,3,64)
