/* c-ray-hip build: stand-in for the file the reference's CMake generates from src/utils/gitsha1.c.in
 * (CMakeLists.txt:69-71); the reference tree carries no git metadata here. */
char *gitHash(void) {
	return "mi355x";
}
