/* Pulls in the C ABI; pass -Xcc -I<repo>/include so that the header is found. */
#include <fluidsolver.h>
