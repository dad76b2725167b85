// Configurations of the persistent pointwise convolution kernel that are compiled in (16-bit types):
// X(BM, KC, PT) = channels of the resident weight panel, padded K of the panel, 16-pixel sub-tiles a wave takes per step.
#pragma once
#define LH_PW_CFGS(X) \
    X(256,64,1) X(128,64,2) X(64,64,2) X(64,64,4) \
    X(256,128,1) X(128,128,2) X(64,128,2) \
    X(256,256,1) X(128,256,1) X(128,256,2) X(64,256,1) X(64,256,2) \
    X(128,512,1) X(64,512,1)
